// lds_occupancy.hip -- how many one-wave blocks fit a CU as a function of their dynamic LDS size (MI355X box):
// hipOccupancyMaxActiveBlocksPerMultiprocessor and the device properties, to size per-wave LDS budgets.
#include <hip/hip_runtime.h>
#include <stdio.h>

extern __shared__ unsigned char smem[];
__global__ void __launch_bounds__(64) probe(unsigned *out) {
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = smem[(threadIdx.x + 1) & 63];
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    printf("sharedMemPerBlock %zu  sharedMemPerMultiprocessor %zu  maxSharedMemoryPerMultiProcessor %zu  regsPerBlock %d  multiProcessorCount %d\n",
           p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.multiProcessorCount);
    for (int kb2 = 24; kb2 <= 44; ++kb2) {            // 12 .. 22 KB in steps of 512 B
        const size_t bytes = (size_t)kb2 * 512;
        int blocks = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, probe, 64, bytes);
        printf("dynamic LDS %6zu B: %d blocks per CU (%s)\n", bytes, blocks, hipGetErrorString(e));
    }
    for (size_t bytes : {(size_t)18688, (size_t)18432, (size_t)17408, (size_t)16384, (size_t)20480, (size_t)20992})
    {
        int blocks = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, probe, 64, bytes);
        printf("dynamic LDS %6zu B: %d blocks per CU\n", bytes, blocks);
    }
    return 0;
}
