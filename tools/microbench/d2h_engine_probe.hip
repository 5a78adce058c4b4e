// d2h_engine_probe.hip -- which engine carries a device-to-host hipMemcpyAsync, depending on what precedes it in the
// stream?  Run under `rocprofv3 --kernel-trace --memory-copy-trace`: an SDMA transfer shows up as a
// MEMORY_COPY_DEVICE_TO_HOST record, a shader copy as a __amd_rocclr_copyBuffer kernel.  Every variant copies a
// different size so that the records can be told apart.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void touch(unsigned *p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1;
}

int main() {
    const size_t MB = 1 << 20;
    void *dev = nullptr, *host = nullptr;
    CK(hipMalloc(&dev, 256 * MB));
    CK(hipHostMalloc(&host, 256 * MB, hipHostMallocDefault));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const unsigned blocks = (unsigned)(64 * MB / 4 / 256);
    // V0 (64 MB): nothing before the copy
    CK(hipMemcpyAsync(host, dev, 64 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V1 (65 MB): a kernel on the same stream right before it
    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, a, (unsigned *)dev, 64 * MB / 4);
    CK(hipMemcpyAsync(host, dev, 65 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V2 (66 MB): a kernel on another stream, event, hipStreamWaitEvent, copy
    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, b, (unsigned *)dev, 64 * MB / 4);
    CK(hipEventRecord(ev, b));
    CK(hipStreamWaitEvent(a, ev, 0));
    CK(hipMemcpyAsync(host, dev, 66 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V3 (67 MB): a kernel on another stream, the HOST waits for the event, then the copy
    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, b, (unsigned *)dev, 64 * MB / 4);
    CK(hipEventRecord(ev, b));
    CK(hipEventSynchronize(ev));
    CK(hipMemcpyAsync(host, dev, 67 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V4 (68 MB): an H2D copy on the same stream before it (no kernel anywhere near)
    CK(hipMemcpyAsync(dev, host, 8 * MB, hipMemcpyHostToDevice, a));
    CK(hipMemcpyAsync(host, dev, 68 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V5 (69 MB): event recorded after a D2H on another stream (copy -> copy dependency)
    CK(hipMemcpyAsync(host, dev, 1 * MB, hipMemcpyDeviceToHost, b));
    CK(hipEventRecord(ev, b));
    CK(hipStreamWaitEvent(a, ev, 0));
    CK(hipMemcpyAsync((char *)host + 128 * MB, dev, 69 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    // V6 (70 MB): kernel on the SAME stream, then hipStreamSynchronize, then the copy
    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, a, (unsigned *)dev, 64 * MB / 4);
    CK(hipStreamSynchronize(a));
    CK(hipMemcpyAsync(host, dev, 70 * MB, hipMemcpyDeviceToHost, a));
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
