// Development experiment: what does a dependent kernel boundary cost on this GPU as a function of the AQL packet
// header — barrier bit, acquire / release fence scope — when the dispatch packets are written by hand into an HSA
// queue instead of going through hipLaunchKernel?  (The HIP runtime picks the header itself; see AMD_LOG_LEVEL=4.)
// Kernel: tools/ubench/aql_kernel.hip, a stand-in for the headline step kernel (512 single-wave workgroups).
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only --no-gpu-bundle-output -c -o /tmp/aql_kernel.hsaco tools/ubench/aql_kernel.hip
//   hipcc -O2 -o /tmp/aql_boundary tools/ubench/aql_boundary.cpp -lhsa-runtime64 && /tmp/aql_boundary /tmp/aql_kernel.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool;
static bool g_have_kernarg = false;
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_kernarg) { g_kernarg_pool = p; g_have_kernarg = true; }
    return HSA_STATUS_SUCCESS;
}

// Interposed in front of libhsa-runtime64: the HIP runtime's own hsa_queue_create calls land here, so that the experiment
// can (USE_HIP_QUEUE=1) write its packets into a queue the HIP runtime created — is a boundary slower in "our" queue
// because of how the queue was made?
#include <dlfcn.h>
#include <hsa/amd_hsa_queue.h>
static hsa_queue_t* g_seen[16]; static int g_nseen = 0;
extern "C" hsa_status_t hsa_queue_create(hsa_agent_t agent, uint32_t size, hsa_queue_type32_t type,
                                         void (*callback)(hsa_status_t, hsa_queue_t*, void*), void* data,
                                         uint32_t private_segment_size, uint32_t group_segment_size, hsa_queue_t** queue) {
    typedef hsa_status_t (*fn_t)(hsa_agent_t, uint32_t, hsa_queue_type32_t, void (*)(hsa_status_t, hsa_queue_t*, void*), void*, uint32_t, uint32_t, hsa_queue_t**);
    static fn_t real = (fn_t)dlsym(RTLD_NEXT, "hsa_queue_create");
    const hsa_status_t st = real(agent, size, type, callback, data, private_segment_size, group_segment_size, queue);
    if (st == HSA_STATUS_SUCCESS && g_nseen < 16) g_seen[g_nseen++] = *queue;
    fprintf(stderr, "[interposed] hsa_queue_create(size %u, type %u, cb %p, priv %u, group %u) -> %p\n", size, (unsigned)type, (void*)callback, private_segment_size, group_segment_size, (void*)*queue);
    return st;
}
static void dump_queue(const char* who, hsa_queue_t* q) {
    const amd_queue_t* a = (const amd_queue_t*)q;
    fprintf(stderr, "[queue %s] %p type %u features %#x size %u base %p props %#x max_cu_id %u max_wave_id %u scratch_wave64_lane_byte_size %u compute_tmpring_size %#x\n",
            who, (void*)q, (unsigned)q->type, q->features, q->size, q->base_address, a->queue_properties, a->max_cu_id, a->max_wave_id, a->scratch_wave64_lane_byte_size, a->compute_tmpring_size);
}

struct Args { float* p; unsigned* bad; int B; int sc1; };

__global__ void dummy_kernel() {}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* path = argc > 1 ? argv[1] : "/tmp/aql_kernel.hsaco";
    const int K = argc > 2 ? atoi(argv[2]) : 3000;   // < queue size: all packets of a leg are queued before the gate opens
    HK(hipSetDevice(0));
    float* p; unsigned* bad;
    const int B = 4096;
    HK(hipMalloc(&p, (size_t)48 * B * 4)); HK(hipMemset(p, 0, (size_t)48 * B * 4));
    HK(hipMalloc(&bad, 8192)); HK(hipMemset(bad, 0, 8192));
    HK(hipDeviceSynchronize());

    CK(hsa_init());
    CK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no agents\n"); return 1; }
    CK(hsa_amd_agent_iterate_memory_pools(g_cpu, pool_cb, nullptr));
    if (!g_have_kernarg) { fprintf(stderr, "no kernarg pool\n"); return 1; }

    // code object
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(sz);
    if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 1;
    fclose(f);
    hsa_code_object_reader_t reader;
    CK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe;
    CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(exe, "rows_chain.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t kasz = 0, lds = 0, scratch = 0;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kasz));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &scratch));
    printf("kernel object %#lx kernarg %u lds %u scratch %u\n", (unsigned long)kobj, kasz, lds, scratch);

    hsa_queue_t* q;
    CK(hsa_queue_create(g_gpu, getenv("QUEUE_SIZE") ? atoi(getenv("QUEUE_SIZE")) : 4096, getenv("QUEUE_MULTI") ? HSA_QUEUE_TYPE_MULTI : HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    dump_queue("ours", q);
    for (int i = 0; i < g_nseen; ++i) dump_queue(g_seen[i] == q ? "ours again" : "seen", g_seen[i]);
    if (getenv("USE_HIP_QUEUE")) {   // the queue of the stream created below does not exist yet: take the newest one HIP made so far
        hipStream_t s0; HK(hipStreamCreate(&s0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dummy_kernel), dim3(1), dim3(64), 0, s0); HK(hipStreamSynchronize(s0));
        hsa_queue_t* hq = nullptr;
        for (int i = g_nseen - 1; i >= 0; --i) if (g_seen[i] != q) { hq = g_seen[i]; break; }
        if (hq) { fprintf(stderr, "writing the packets into HIP's queue %p\n", (void*)hq); q = hq; dump_queue("hip", q); }
    }
    if (getenv("QUEUE_PROFILING")) CK(hsa_amd_profiling_set_profiler_enabled(q, 1));
    hsa_signal_t done;
    CK(hsa_signal_create(1, 0, nullptr, &done));

    // one kernarg block per variant of sc1 (contents never change between launches)
    // kernel arguments in DEVICE memory, like the HIP runtime's default on this GPU (host-memory kernargs cost a PCIe
    // round trip per wave: +3 us per launch); KERNARG_HOST=1 puts them into the CPU agent's kernarg pool instead
    Args* ka;
    if (getenv("KERNARG_HOST")) {
        CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 4096, 0, (void**)&ka));
        CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka));
        ka[0] = Args{p, bad, B, 0};
        ka[8] = Args{p, bad, B, 1};   // 8 * 24 B apart
    } else {
        HK(hipMalloc((void**)&ka, 1 << 20));
        std::vector<Args> h((1 << 20) / sizeof(Args));
        for (size_t i = 0; i < h.size(); ++i) h[i] = Args{p, bad, B, (i % 16) == 8 ? 1 : 0};   // every 384-byte slot starts a copy
        HK(hipMemcpy(ka, h.data(), 1 << 20, hipMemcpyHostToDevice));
        HK(hipDeviceSynchronize());
    }

    void* hip_kernarg = nullptr;
    auto submit = [&](uint16_t header, void* kernarg, hsa_signal_t sig) {
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
        hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
        pk->workgroup_size_x = 64; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1;
        pk->reserved0 = 0;
        pk->grid_size_x = 512 * 64; pk->grid_size_y = 1; pk->grid_size_z = 1;
        pk->private_segment_size = scratch; pk->group_segment_size = lds;
        pk->kernel_object = kobj; pk->kernarg_address = hip_kernarg ? hip_kernarg : kernarg; pk->reserved2 = 0;
        pk->completion_signal = sig;
        const uint16_t setup = (getenv("SETUP3") ? 3 : 1) << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n((uint32_t*)pk, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, idx);
    };
    auto hdr = [](int barrier, int acq, int rel) -> uint16_t {
        return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    };
    const hsa_signal_t none = {0};
    const uint16_t full = getenv("FULL_AGENT") ? hdr(1, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT) : hdr(1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM);
    // All K packets are written while the queue is held by a barrier-AND packet that waits for `gate`; the clock runs
    // from the opening of the gate to the completion signal of the last packet: GPU-side processing only, no host
    // submission in the timed region.
    hsa_signal_t gate;
    CK(hsa_signal_create(1, 0, nullptr, &gate));
    auto hold = [&]() {
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        hsa_barrier_and_packet_t* pk = (hsa_barrier_and_packet_t*)q->base_address + (idx & (q->size - 1));
        memset((char*)pk + 4, 0, sizeof(*pk) - 4);
        pk->dep_signal[0] = gate;
        const uint16_t h = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER));
        __atomic_store_n((uint32_t*)pk, (uint32_t)h, __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, idx);
    };
    hsa_signal_t dummy;   // EVERY_SIGNAL=1: every packet carries a completion signal (as the HIP runtime's do)
    CK(hsa_signal_create(1 << 30, 0, nullptr, &dummy));
    const bool every = getenv("EVERY_SIGNAL") != nullptr;
    const bool rot = getenv("ROTATE_KERNARG") != nullptr && !getenv("KERNARG_HOST");
    if (hip_kernarg) printf("using the HIP runtime's kernarg block %p for every packet\n", hip_kernarg);
    auto run = [&](const char* name, uint16_t h, int sc1) {
        for (int rep = 0; rep < 2; ++rep) {
            hsa_signal_store_relaxed(done, 1);
            hsa_signal_store_relaxed(gate, 1);
            auto ts = std::chrono::steady_clock::now();
            const int depth = getenv("DEPTH") ? atoi(getenv("DEPTH")) : 0;   // > 0: no gate, at most DEPTH packets ahead of the read index
            if (depth) {
                auto t1 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; ++k) {
                    while ((int64_t)(hsa_queue_load_write_index_relaxed(q) - hsa_queue_load_read_index_scacquire(q)) >= depth) {}
                    submit(k == 0 || k == K - 1 ? full : h, &ka[sc1 ? 8 : 0], k == K - 1 ? done : none);
                }
                while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
                const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / K;
                if (rep) printf("%-58s %7.3f us per launch (at most %d packets queued)\n", name, us1, depth);
                continue;
            }
            hold();
            submit(full, &ka[sc1 ? 8 : 0], none);
            for (int k = 0; k < K - 2; ++k) submit(h, &ka[(sc1 ? 8 : 0) + (rot ? 16 * (k % 2000) : 0)], every ? dummy : none);
            submit(full, &ka[sc1 ? 8 : 0], done);
            const double sub_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ts).count() / K;
            auto t0 = std::chrono::steady_clock::now();
            hsa_signal_store_screlease(gate, 0);
            while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            if (rep) printf("%-58s %7.3f us per launch (host: %.2f us per packet)\n", name, us, sub_us);
        }
    };
    // the same kernel through HIP for reference
    {
        hipModule_t mod; hipFunction_t fn;
        HK(hipModuleLoadData(&mod, blob.data()));
        HK(hipModuleGetFunction(&fn, mod, "rows_chain"));
        hipStream_t s; HK(hipStreamCreate(&s));
        Args a{p, bad, B, 0};
        size_t asz = sizeof(a);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        if (getenv("DUMP_PENDING")) {   // the HIP runtime's packets BEFORE the packet processor consumes (and invalidates) them
            for (int k = 0; k < 200; ++k) HK(hipModuleLaunchKernel(fn, 512, 1, 1, 64, 1, 1, 0, s, nullptr, cfg));
            for (int i = 0; i < g_nseen; ++i) {
                hsa_queue_t* hq = g_seen[i];
                const uint64_t w = hsa_queue_load_write_index_relaxed(hq), r = hsa_queue_load_read_index_relaxed(hq);
                fprintf(stderr, "[queue %p: read index %lu, write index %lu]\n", (void*)hq, (unsigned long)r, (unsigned long)w);
                for (uint64_t k = (w > 4 ? w - 4 : 0); k < w; ++k) {
                    const uint32_t* d = (const uint32_t*)((const char*)hq->base_address + 64 * (k & (hq->size - 1)));
                    fprintf(stderr, "  #%lu%s:", (unsigned long)k, k >= r ? " (pending)" : "");
                    for (int j = 0; j < 16; ++j) fprintf(stderr, " %08x", d[j]);
                    fprintf(stderr, "\n");
                }
            }
            HK(hipStreamSynchronize(s));
        }
        for (int rep = 0; rep < 2; ++rep) {
            HK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) HK(hipModuleLaunchKernel(fn, 512, 1, 1, 64, 1, 1, 0, s, nullptr, cfg));
            HK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            if (rep) printf("%-58s %7.3f us per launch\n", "hipModuleLaunchKernel, one stream", us);
        }
    }
    auto xcc_log = [&](const char* who) {
        unsigned w[68];
        HK(hipMemcpy(w, bad + 1100, sizeof(w), hipMemcpyDeviceToHost));
        printf("  XCC of block 0 in the last launches (%s, ring of 64, %u launches so far):", who, w[0]);
        for (int i = 0; i < 24; ++i) printf(" %u", w[4 + ((w[0] - 24 + i) & 63u)]);
        printf("\n");
    };
    auto placement = [&](const char* who) {   // HW_ID: wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13]
        std::vector<unsigned> w(2 * 512 + 32);
        HK(hipMemcpy(w.data(), bad, w.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int> per(8 * 8 * 2 * 16 * 4, 0);
        int cus = 0, simds = 0, mx = 0;
        std::vector<int> cu_used(8 * 8 * 2 * 16, 0);
        for (int b = 0; b < 512; ++b) {
            const unsigned h = w[32 + 2 * b], x = w[33 + 2 * b] & 7u;
            const unsigned simd = (h >> 4) & 3u, cu = (h >> 8) & 15u, sh = (h >> 12) & 1u, se = (h >> 13) & 7u;
            const int ci = (int)(((x * 8 + se) * 2 + sh) * 16 + cu);
            if (!cu_used[ci]++) ++cus;
            if (!per[ci * 4 + simd]++) ++simds;
            if (per[ci * 4 + simd] > mx) mx = per[ci * 4 + simd];
        }
        printf("  placement (%s): 512 waves on %d CUs, %d SIMDs, at most %d waves on one SIMD; block 0..7 on XCC", who, cus, simds, mx);
        for (int b = 0; b < 8; ++b) printf(" %u", w[33 + 2 * b] & 7u);
        printf("\n");
    };
    placement("last HIP launch");
    xcc_log("HIP launches");
    if (getenv("USE_HIP_KOBJ")) {   // dispatch the copy of the kernel the HIP runtime loaded: its kernel object is in the packets it wrote
        for (int i = 0; i < g_nseen; ++i) {
            hsa_queue_t* hq = g_seen[i];
            const uint64_t w = hsa_queue_load_write_index_relaxed(hq);
            for (uint64_t k = w > 8 ? w - 8 : 0; k < w; ++k) {
                const hsa_kernel_dispatch_packet_t* d = (const hsa_kernel_dispatch_packet_t*)hq->base_address + (k & (hq->size - 1));
                if (d->workgroup_size_x == 64 && d->grid_size_x == 32768 && d->kernel_object) {
                    kobj = d->kernel_object;
                    if (getenv("USE_HIP_KERNARG")) { hip_kernarg = d->kernarg_address; }
                }
            }
        }
        printf("using the HIP runtime's kernel object %#lx\n", (unsigned long)kobj);
    }
    if (getenv("DUMP_PACKETS")) {   // what do the HIP runtime's own dispatch packets look like, byte for byte?
        for (int i = 0; i < g_nseen; ++i) {
            hsa_queue_t* hq = g_seen[i];
            const uint64_t w = hsa_queue_load_write_index_relaxed(hq);
            fprintf(stderr, "[packets of queue %p, write index %lu]\n", (void*)hq, (unsigned long)w);
            for (uint64_t k = w > 3 ? w - 3 : 0; k < w; ++k) {
                const uint32_t* d = (const uint32_t*)((const char*)hq->base_address + 64 * (k & (hq->size - 1)));
                fprintf(stderr, "  #%lu:", (unsigned long)k);
                for (int j = 0; j < 16; ++j) fprintf(stderr, " %08x", d[j]);
                fprintf(stderr, "\n");
            }
        }
    }
    const int A = HSA_FENCE_SCOPE_AGENT, S = HSA_FENCE_SCOPE_SYSTEM, N = HSA_FENCE_SCOPE_NONE;
    if (getenv("LATE_QUEUE")) {   // a queue made AFTER the HIP runtime has created its stream's queue and run kernels
        CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
        printf("the packets below go to a queue created after the HIP launches: %p\n", (void*)q);
    }
    if (getenv("MANY_QUEUES")) {   // is the cost of a barrier a property of the QUEUE (which hardware pipe / slot it got)?
        hsa_queue_t* q0 = q;
        for (int i = 0; i < atoi(getenv("MANY_QUEUES")); ++i) {
            hsa_queue_t* qi;
            CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qi));
            q = qi;
            char name[96];
            snprintf(name, sizeof name, "queue #%d (%p) barrier=1 agent/agent", i, (void*)qi);
            run(name, hdr(1, A, A), 0);
            xcc_log(name);
        }
        q = q0;
    }
    run("AQL barrier=1 acquire=system release=system", hdr(1, S, S), 0);
    run("AQL barrier=1 acquire=agent  release=agent", hdr(1, A, A), 0);
    placement("last AQL launch");
    xcc_log("AQL barrier=1 agent/agent");
    run("AQL barrier=1 acquire=agent  release=none", hdr(1, A, N), 0);
    run("AQL barrier=1 acquire=none   release=agent", hdr(1, N, A), 0);
    run("AQL barrier=1 acquire=none   release=none", hdr(1, N, N), 0);
    run("AQL barrier=1 acquire=none   release=none, nt loads", hdr(1, N, N), 1);
    run("AQL barrier=0 acquire=none   release=none (may overlap)", hdr(0, N, N), 0);
    run("AQL barrier=0 acquire=agent  release=agent (may overlap)", hdr(0, A, A), 0);
    xcc_log("AQL barrier=0");
    unsigned nbad = 0;
    HK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
    float first = 0;
    HK(hipMemcpy(&first, p, 4, hipMemcpyDeviceToHost));
    printf("blocks not on XCD (blockIdx %% 8): %u; row0[0] = %.0f (launches that incremented it)\n", nbad, first);
    unsigned map[16];
    HK(hipMemcpy(map, bad + 1, sizeof(map), hipMemcpyDeviceToHost));
    printf("XCC_ID register of blocks 0..15 in the last launch:");
    for (int i = 0; i < 16; ++i) printf(" %#x", map[i]);
    printf("\n");
    if (!getenv("USE_HIP_QUEUE")) hsa_queue_destroy(q);
    return 0;
}
