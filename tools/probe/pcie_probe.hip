// Host <-> device transfer rates on the GPU box (pageable, registered in place, pinned ring + threaded staging copies):
// the numbers behind the staging design of bftObj_bftBatch (afx_bft.c).  hipcc -O2 -o pcie_probe pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char *d, const char *s, size_t n, int threads) {
    std::vector<std::thread> t;
    const size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int i = 0; i < threads; i++) {
        const size_t lo = per * i, hi = lo + per < n ? lo + per : n;
        if (lo < hi) t.emplace_back([=] { memcpy(d + lo, s + lo, hi - lo); });
    }
    for (auto &x : t) x.join();
}
int main() {
    const size_t IN = 200ull * 480000 * 4, OUT = 200ull * 934 * 128 * 4;
    printf("host threads: %u\n", std::thread::hardware_concurrency());
    char *pin = (char *)malloc(IN), *pout = (char *)malloc(OUT), *din, *dout, *hin, *hout;
    memset(pin, 1, IN);
    memset(pout, 1, OUT);
    CK(hipMalloc(&din, IN));
    CK(hipMalloc(&dout, OUT));
    CK(hipHostMalloc(&hin, IN, hipHostMallocDefault));
    CK(hipHostMalloc(&hout, OUT, hipHostMallocDefault));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto small = [&](const char *what) {
        double t0 = now();
        for (int k = 0; k < 100; k++) {
            (void)hipMemcpyAsync(din, pin + (size_t)k * 1920000, 1920000, hipMemcpyHostToDevice, s1);
            (void)hipStreamSynchronize(s1);
        }
        double t1 = now();
        for (int k = 0; k < 100; k++) {
            (void)hipMemcpyAsync(pout + (size_t)k * 478208, dout, 478208, hipMemcpyDeviceToHost, s1);
            (void)hipStreamSynchronize(s1);
        }
        double t2 = now();
        printf("100 pageable copies of one clip (%s): H2D %.1f us each (%.1f GB/s), D2H of its mel rows %.1f us each\n", what,
               (t1 - t0) * 1e4, 1.92e6 / ((t1 - t0) / 100) / 1e9, (t2 - t1) * 1e4);
        return 0;
    };
    small("fresh buffers");
    small("fresh buffers, again");
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipMemcpy(din, pin, IN, hipMemcpyHostToDevice));
        double t1 = now();
        CK(hipMemcpy(pout, dout, OUT, hipMemcpyDeviceToHost));
        double t2 = now();
        printf("pageable   H2D %.1f GB/s   D2H %.1f GB/s\n", IN / (t1 - t0) / 1e9, OUT / (t2 - t1) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(din, hin, IN, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        t1 = now();
        CK(hipMemcpyAsync(hout, dout, OUT, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s2));
        t2 = now();
        printf("pinned     H2D %.1f GB/s   D2H %.1f GB/s\n", IN / (t1 - t0) / 1e9, OUT / (t2 - t1) / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(din, hin, IN, hipMemcpyHostToDevice, s1));
        for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(hout, dout, OUT, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1));
        t1 = now();
        CK(hipStreamSynchronize(s2));
        t2 = now();
        printf("pinned both ways at once: H2D %.1f GB/s, D2H (4 x) %.1f GB/s\n", IN / (t1 - t0) / 1e9, 4 * OUT / (t2 - t0) / 1e9);
        t0 = now();
        CK(hipHostRegister(pin, IN, hipHostRegisterDefault));
        t1 = now();
        CK(hipMemcpyAsync(din, pin, IN, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        t2 = now();
        CK(hipHostUnregister(pin));
        double t3 = now();
        printf("register in place: %.2f ms (%.1f GB/s), H2D %.1f GB/s, unregister %.2f ms -> end to end %.1f GB/s\n", (t1 - t0) * 1e3,
               IN / (t1 - t0) / 1e9, IN / (t2 - t1) / 1e9, (t3 - t2) * 1e3, IN / (t3 - t0) / 1e9);
        small("after register / unregister of the whole buffer");
        for (int th : {1, 2, 4, 8, 16}) {
            t0 = now();
            par_copy(hin, pin, IN, th);
            t1 = now();
            par_copy(pout, hout, OUT, th);
            t2 = now();
            printf("memcpy pageable -> pinned, %2d threads: %.1f GB/s;  pinned -> pageable: %.1f GB/s\n", th, IN / (t1 - t0) / 1e9, OUT / (t2 - t1) / 1e9);
        }
        // two host threads issuing pageable copies on two streams at once
        t0 = now();
        std::thread up([&] { (void)hipMemcpyAsync(din, pin, IN, hipMemcpyHostToDevice, s1); (void)hipStreamSynchronize(s1); });
        for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(pout, dout, OUT, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s2));
        t2 = now();
        up.join();
        t1 = now();
        printf("pageable both ways from two host threads: H2D %.1f GB/s, D2H (4 x) %.1f GB/s\n", IN / (t1 - t0) / 1e9, 4 * OUT / (t2 - t0) / 1e9);
    }
    return 0;
}
