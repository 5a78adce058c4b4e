// d2h_probe.hip -- what the device's copy engine delivers into host memory of different kinds (MI355X box):
// hipHostMalloc'ed staging vs malloc'ed memory registered with hipHostRegister (touched / untouched), whole and in
// 128 MB chunks, alone and with an H2D stream running beside it.  Decides how valign_hip_align_host should deliver
// 1.36 GB of result rows per million pairs (hip_engine.hip.h: align_host).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = (size_t)1363148800;          // 1,048,576 pairs x 2 x 650
    const size_t chunk = (size_t)128 << 20;
    void *dev = nullptr, *dev_in = nullptr, *pinned = nullptr, *pinned_in = nullptr;
    CK(hipMalloc(&dev, bytes));
    CK(hipMalloc(&dev_in, bytes / 2));
    CK(hipMemset(dev, 1, bytes));
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&pinned_in, bytes / 2, hipHostMallocDefault));
    memset(pinned_in, 2, bytes / 2);
    char *plain = (char *)malloc(bytes + 4096), *fresh = (char *)malloc(bytes + 4096);
    memset(plain, 0, bytes);                          // touched; `fresh` stays untouched until registered
    hipStream_t s_out, s_in;
    CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    double t0 = now_ms();
    CK(hipHostRegister(plain, bytes, hipHostRegisterDefault));
    printf("hipHostRegister(touched 1.36 GB): %.1f ms\n", now_ms() - t0);
    t0 = now_ms();
    CK(hipHostRegister(fresh, bytes, hipHostRegisterDefault));
    printf("hipHostRegister(untouched 1.36 GB): %.1f ms\n", now_ms() - t0);
    struct { const char *name; void *dst; } kinds[] = {{"hipHostMalloc", pinned}, {"registered (touched)", plain}, {"registered (was untouched)", fresh}};
    for (int duplex = 0; duplex < 2; ++duplex)
        for (auto &k : kinds)
            for (int chunked = 0; chunked < 2; ++chunked)
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    t0 = now_ms();
                    if (duplex) CK(hipMemcpyAsync(dev_in, pinned_in, bytes / 2, hipMemcpyHostToDevice, s_in));
                    if (chunked) {
                        for (size_t at = 0; at < bytes; at += chunk)
                            CK(hipMemcpyAsync((char *)k.dst + at, (char *)dev + at, bytes - at < chunk ? bytes - at : chunk, hipMemcpyDeviceToHost, s_out));
                    } else {
                        CK(hipMemcpyAsync(k.dst, dev, bytes, hipMemcpyDeviceToHost, s_out));
                    }
                    CK(hipStreamSynchronize(s_out));
                    const double ms = now_ms() - t0;
                    CK(hipStreamSynchronize(s_in));
                    if (rep == 2) printf("D2H 1.36 GB -> %-28s %s%s: %.2f ms = %.1f GB/s\n", k.name, chunked ? "128 MB chunks" : "one copy", duplex ? " + H2D 0.68 GB beside it" : "", ms, bytes / ms / 1e6);
                }
    // host-side copy out of the pinned staging (what the staged path adds): 16 threads would split this
    t0 = now_ms();
    memcpy(fresh, pinned, bytes);
    printf("memcpy 1.36 GB pinned -> malloc'ed, one thread: %.1f ms\n", now_ms() - t0);
    return 0;
}
