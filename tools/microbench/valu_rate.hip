// Issue-rate microbenchmark for the packed-int16 VALU ops the DP kernels are made of.
// Prints wave-instructions per cycle per SIMD for 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define ITER 4096
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(512) rate(unsigned *out, unsigned seed, long long *cycles) {
    unsigned r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r[i] = seed * (i + 1) + threadIdx.x;
    const unsigned g = seed | 0x00030003u;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            s16x2 a = __builtin_bit_cast(s16x2, r[i]);
            s16x2 b = __builtin_bit_cast(s16x2, g);
            if (OP == 0) a = __builtin_elementwise_max(a, b);                                   // v_pk_max_i16
            if (OP == 1) a = a + b;                                                               // v_pk_add_u16
            if (OP == 2) a = (s16x2)__builtin_elementwise_sub_sat((u16x2)a, (u16x2)b);           // v_pk_sub_u16 clamp
            if (OP == 3) r[i] = __builtin_amdgcn_perm(r[i], g, 0x05040100u ^ r[(i + 1) % UNROLL]); // v_perm_b32 (+xor)
            if (OP == 4) r[i] = r[i] + g;                                                         // v_add_u32
            if (OP == 5) r[i] = (unsigned)max((int)r[i], (int)g);                                 // v_max_i32
            if (OP == 6) a = __builtin_elementwise_add_sat(a, b);                                 // v_pk_add_i16 clamp
            if (OP == 7) {                                                                          // v_pk_add_f16
                f16x2 x = __builtin_bit_cast(f16x2, r[i]) + __builtin_bit_cast(f16x2, g);
                r[i] = __builtin_bit_cast(unsigned, x);
            }
            if (OP == 8) {                                                                          // v_pk_maximum3_f16
                f16x2 x = __builtin_elementwise_maximum(
                    __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, r[i]), __builtin_bit_cast(f16x2, g)),
                    __builtin_bit_cast(f16x2, r[(i + 1) % UNROLL]));
                r[i] = __builtin_bit_cast(unsigned, x);
            }
            if (OP == 9) {                                                                          // 2-operand maximum
                f16x2 x = __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, r[i]), __builtin_bit_cast(f16x2, g));
                r[i] = __builtin_bit_cast(unsigned, x);
            }
            if (OP != 3 && OP != 4 && OP != 5 && OP < 7) r[i] = __builtin_bit_cast(unsigned, a);
            asm volatile("" : "+v"(r[i]));
        }
    }
    long long t1 = clock64();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name) {
    unsigned *out;
    long long *cyc;
    hipMalloc(&out, 256 * 512 * 8 * sizeof(unsigned));
    hipMalloc(&cyc, 4096 * sizeof(long long));
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd);   // block = up to 8 waves
        const int blocks_per_cu = waves_per_simd > 2 ? waves_per_simd / 2 : 1;
        const int blocks = 256 * blocks_per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        rate<OP><<<blocks, threads>>>(out, 12345u, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<OP><<<blocks, threads>>>(out, 12345u, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= blocks;
        const double instr_per_wave = (double)ITER * UNROLL * (OP == 3 ? 2 : 1);
        // s_memtime/clock64 ticks at 100 MHz on gfx9: use wall time + nominal clock instead
        const double wave_instr = instr_per_wave * blocks * (threads / 64);
        const double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
        printf("%-22s waves/SIMD %d: %.3f ms, %.3f G wave-instr/s/SIMD  (= %.2f cycles/instr at 2.4 GHz)\n", name,
               waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_pk_max_i16");
    run<1>("v_pk_add_u16");
    run<2>("v_pk_sub_u16 clamp");
    run<6>("v_pk_add_i16 clamp");
    run<3>("v_perm_b32 + v_xor");
    run<4>("v_add_u32");
    run<5>("v_max_i32");
    run<7>("v_pk_add_f16");
    run<8>("v_pk_maximum3_f16");
    run<9>("f16x2 maximum(a, b)");
    return 0;
}
