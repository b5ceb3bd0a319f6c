// rsx_launch.hpp — kernel launches whose status is THIS library's own.
//
// hipLaunchKernelGGL reports a failed launch only through the thread's last-error slot, which every HIP user of the process shares:
// checking it after a launch picks up whatever a caller's earlier work left there (torch ending an aborted stream capture leaves
// `invalid argument`), and clearing it on entry — what this library did in ABI 5 — silently swallows the caller's error.  rsx_launch
// goes through hipLaunchKernel, whose return value IS the launch's status, and keeps the first failure of the current API call in a
// thread-local of its own (launch_status(): read and reset).  The shared slot is neither read nor cleared on the stepping paths.
#pragma once
#include <hip/hip_runtime.h>

#include <tuple>
#include <utility>

namespace rsx {

inline thread_local hipError_t g_launch_status = hipSuccess;

// the first launch error since the last call (hipSuccess if none); resets the record
inline hipError_t launch_status() {
    const hipError_t e = g_launch_status;
    g_launch_status = hipSuccess;
    return e;
}

template <typename... KArgs, size_t... I>
inline hipError_t launch_impl(void (*kernel)(KArgs...), const dim3 grid, const dim3 block, const size_t shmem, hipStream_t s,
                              std::tuple<KArgs...>& vals, std::index_sequence<I...>) {
    void* ptrs[] = {(void*)&std::get<I>(vals)..., nullptr};
    return hipLaunchKernel((const void*)kernel, grid, block, ptrs, shmem, s);
}

// arguments are converted to the kernel's own parameter types (what the <<<>>> syntax does implicitly)
template <typename... KArgs, typename... Args>
inline void rsx_launch(void (*kernel)(KArgs...), const dim3 grid, const dim3 block, const size_t shmem, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count does not match the kernel's parameter list");
    std::tuple<KArgs...> vals{static_cast<KArgs>(std::forward<Args>(args))...};
    const hipError_t e = launch_impl(kernel, grid, block, shmem, s, vals, std::index_sequence_for<KArgs...>{});
    if (e != hipSuccess && g_launch_status == hipSuccess) g_launch_status = e;
}

}  // namespace rsx
