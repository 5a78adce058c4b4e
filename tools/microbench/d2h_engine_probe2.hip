// d2h_engine_probe2.hip -- second round of tools/microbench/d2h_engine_probe.hip: is a D2H hipMemcpyAsync handed to
// the SDMA engine or to a shader when the GPU is BUSY at the moment the copy is issued?  (rocprofv3 --kernel-trace
// --memory-copy-trace: SDMA = MEMORY_COPY_DEVICE_TO_HOST record, shader = __amd_rocclr_copyBuffer kernel.)
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <thread>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned *p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) p[blockIdx.x] += 1;
}

int main() {
    const size_t MB = 1 << 20;
    void *dev = nullptr, *host = nullptr, *dev2 = nullptr, *host2 = nullptr;
    CK(hipMalloc(&dev, 256 * MB));
    CK(hipMalloc(&dev2, 256 * MB));
    CK(hipHostMalloc(&host, 256 * MB, hipHostMallocDefault));
    CK(hipHostMalloc(&host2, 256 * MB, hipHostMallocDefault));
    hipStream_t a, b, c, d, s5;
    for (hipStream_t *s : {&a, &b, &c, &d, &s5}) CK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    const long long ms20 = 20ll * 100000;        // clock64 ticks at 100 MHz: ~20 ms
    // V7 (71 MB): a long kernel running on another stream when the copy is issued
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    CK(hipMemcpyAsync(host, dev, 71 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V8 (72 MB): ... and an H2D copy in flight on a third stream
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    CK(hipMemcpyAsync(dev2, host2, 200 * MB, hipMemcpyHostToDevice, c));
    CK(hipMemcpyAsync(host, dev, 72 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V9 (73 MB): issued from another host thread while the long kernel runs
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    std::thread([&] { (void)hipSetDevice(0); (void)hipMemcpyAsync(host, dev, 73 * MB, hipMemcpyDeviceToHost, a); }).join();
    CK(hipDeviceSynchronize());
    // V10 (74 MB + 1 MB): two D2H copies back to back, long kernel running, H2D in flight, D2H in flight on stream d
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    CK(hipMemcpyAsync(dev2, host2, 200 * MB, hipMemcpyHostToDevice, c));
    CK(hipMemcpyAsync((char *)host2 + 200 * MB, (char *)dev2 + 200 * MB, 50 * MB, hipMemcpyDeviceToHost, d));
    CK(hipMemcpyAsync(host, dev, 74 * MB, hipMemcpyDeviceToHost, a));
    CK(hipMemcpyAsync((char *)host + 128 * MB, dev, 1 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V11 (75 MB): a memset on the copy's source on another stream just before (hipMemsetAsync is a fill kernel)
    CK(hipMemsetAsync(dev, 0, 75 * MB, s5));
    CK(hipStreamSynchronize(s5));
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    CK(hipMemcpyAsync(host, dev, 75 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V12 (89,386,700 bytes: 68,759 rows of 1,300 -- a chunk of align_host): odd size
    CK(hipMemcpyAsync(host, dev, (size_t)68759 * 1300, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V13: ... at an odd destination offset
    CK(hipMemcpyAsync((char *)host + 1300 * 7, dev, (size_t)68759 * 1300, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V14 (76 MB): behind a hipStreamWaitEvent on an event of a kernel that is STILL RUNNING
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, b, (unsigned *)dev2, ms20);
    CK(hipEventRecord(ev, b));
    CK(hipStreamWaitEvent(a, ev, 0));
    CK(hipMemcpyAsync(host, dev, 76 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V15 (77 MB): the source was allocated with room to spare and the copy starts inside it
    CK(hipMemcpyAsync(host, (char *)dev + 4096 + 1300, 77 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
