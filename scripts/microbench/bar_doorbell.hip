// Feasibility probe: can the host store straight into device memory (large BAR), and how long does a waiting kernel take to see it?
// Compares: (a) record in pinned HOST memory polled by the device over PCIe (what k_gate does), (b) record in fine-grained DEVICE memory
// written by the host, polled locally.  Prints the round trip host store -> device sees it -> device answers in pinned memory -> host sees it.
// build: hipcc -O2 --offload-arch=gfx950 scripts/microbench/bar_doorbell.hip -o /tmp/bar_doorbell
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void echo(const volatile unsigned long long *door, volatile unsigned long long *answer, int rounds) {
    unsigned long long want = 1;
    for (int r = 0; r < rounds; ++r, ++want) {
        unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
            if (wall_clock64() - t0 > 200000000ull) return;      // 2 s: give up
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(answer, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double run(const char *name, unsigned long long *door_host_view, unsigned long long *door_dev, unsigned long long *ans_h, unsigned long long *ans_d, int rounds) {
    *ans_h = 0;
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, 0, door_dev, ans_d, rounds);
    std::vector<double> rt;
    for (int r = 1; r <= rounds; ++r) {
        for (volatile int spin = 0; spin < 20000; ++spin) {}
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(door_host_view, (unsigned long long)r, __ATOMIC_RELEASE);
        while (__atomic_load_n(ans_h, __ATOMIC_ACQUIRE) != (unsigned long long)r) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3.0) { printf("%s: no answer\n", name); (void)hipDeviceSynchronize(); return -1; }
        }
        rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    (void)hipDeviceSynchronize();
    std::sort(rt.begin(), rt.end());
    printf("%-44s round trip median %.2f us, p10 %.2f, p90 %.2f\n", name, rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10]);
    return rt[rt.size() / 2];
}
int main() {
    unsigned long long *ans_h = nullptr, *ans_d = nullptr, *door_h = nullptr, *door_hd = nullptr;
    hipHostMalloc((void **)&ans_h, 128, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer((void **)&ans_d, ans_h, 0);
    hipHostMalloc((void **)&door_h, 128, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer((void **)&door_hd, door_h, 0);
    *door_h = 0;
    run("door in pinned host memory (k_gate today)", door_h, door_hd, ans_h, ans_d, 2000);
    unsigned long long *door_dev = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&door_dev, 128, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    hipMemset(door_dev, 0, 128); hipDeviceSynchronize();
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
        __atomic_store_n(door_dev, 0ull, __ATOMIC_RELEASE);        // the host touches device memory
        printf("host store into device memory: ok\n");
        run("door in fine-grained device memory", door_dev, door_dev, ans_h, ans_d, 2000);
    } else {
        printf("host store into device memory: fault (no CPU mapping of VRAM)\n");
    }
    return 0;
}
