"""The VSS single-step kernels touch the 8 KB of their own code behind an `s_getpc_b64` with one data load (64 lanes x 128 bytes: every
launch starts with a cold instruction cache, rsx_kernels.hpp: CODE_PF).  A data load from the text segment must stay inside mapped
memory: this test takes the device code objects out of librsx_hip.so and checks, for every kernel that carries the load, that the
window [pc, pc + 8 KB) ends inside the code object's .text section — wherever the linker happened to put the kernel."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "rsoccer_amd", "librsx_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
WINDOW = 64 * 128


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def _code_objects(work):
    """the gfx950 code objects bundled in the library's .hip_fatbin section (one bundle per translation unit)"""
    fat = os.path.join(work, "fat.bin")
    subprocess.check_call([_tool("llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", SO, os.path.join(work, "copy.so")])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    offs = [m.start() for m in re.finditer(re.escape(magic), data)]
    out = []
    for k, o in enumerate(offs):
        b = os.path.join(work, f"bundle{k}.bin")
        open(b, "wb").write(data[o: offs[k + 1] if k + 1 < len(offs) else len(data)])
        co = os.path.join(work, f"co{k}.o")
        subprocess.check_call([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--input={b}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        out.append(co)
    return out


@pytest.mark.skipif(not all(_tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump")),
                    reason="needs the ROCm LLVM binutils")
def test_code_prefetch_window_of_every_kernel_stays_inside_the_text_section():
    work = tempfile.mkdtemp(prefix="rsx_co_")
    try:
        carriers = 0
        for co in _code_objects(work):
            sec = subprocess.check_output([_tool("llvm-readelf"), "-SW", co], text=True)
            m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+[0-9a-f]+\s+([0-9a-f]+)", sec)
            text_lo, text_hi = int(m.group(1), 16), int(m.group(1), 16) + int(m.group(2), 16)
            dis = subprocess.check_output([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
            sym = None
            for line in dis.splitlines():
                h = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if h:
                    sym = h.group(2)
                    continue
                if "s_getpc_b64" in line and sym and "task_step_kernel" in sym:
                    addr = int(re.search(r"//\s*([0-9A-Fa-f]+):", line).group(1), 16) if "//" in line else int(line.split(":")[0], 16)
                    pc = addr + 4          # s_getpc_b64 returns the address of the NEXT instruction
                    carriers += 1
                    assert text_lo <= pc and pc + WINDOW <= text_hi, (sym, hex(pc), hex(text_hi))
                    # only the VSS (KIND 0) single-step (MODE 0) variants carry the load
                    assert re.search(r"task_step_kernelILi0ELi\d+ELi\d+ELi\d+ELi0E", sym), sym
        assert carriers >= 3, carriers     # 3v3 (8 lanes), 5v5 (16 lanes) and the run-time-count variants
    finally:
        shutil.rmtree(work, ignore_errors=True)
