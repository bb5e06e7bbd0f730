// valu_rates.hip — issue rates of the vector instructions the hash phase of k_sketch_tiles is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/gpu/valu_rates.hip && /tmp/valu_rates
// Every kernel runs N_ITER x 8 independent chains x UNROLL dependent instructions of ONE kind per lane, enough waves to fill
// every SIMD (8 per SIMD); reported: wave-instructions per SIMD per cycle-at-2.4-GHz, i.e. 0.25 = one per quad-cycle (full rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;

#define N_ITER 8192
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32 *out, u32 seed, u32 c_in) {
    u32 a[CHAINS], b[CHAINS];
    u64 w[CHAINS];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { a[i] = t * 2654435761u + seed + i; b[i] = a[i] ^ 0x9e3779b9u; w[i] = ((u64)a[i] << 32) | b[i]; }
    const u32 c = c_in | 1u; // (uniform, unknown to the compiler)
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "s"(c) : "vcc");
            if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
            if (OP == 5) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
            if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
            if (OP == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 9) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(c) : "vcc");
            if (OP == 11) asm volatile("v_lshrrev_b64 %0, 31, %0" : "+v"(w[i]));
            if (OP == 12) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
            if (OP == 13) asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_xor_b32 %1, %1, %0" : "+v"(a[i]), "+v"(b[i]) : "s"(c)); // mul + plain, interleaved
            if (OP == 14) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
            if (OP == 15) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % CHAINS]));
            if (OP == 17) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
            if (OP == 18) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
            if (OP == 19) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 20) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 21) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 22) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
            if (OP == 23) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(b[(i + 1) % CHAINS]) : : "vcc");
            if (OP == 24) asm volatile("v_xor_b32_e32 %0, 0x12345678, %0" : "+v"(a[i])); // VOP2 + 32-bit literal (8 bytes)
            if (OP == 26) asm volatile("v_xor_b32_e32 %0, %0, %1\n\tv_mul_lo_u32 %2, %2, %3\n\tv_xor_b32_e32 %1, %1, %0" : "+v"(a[i]), "+v"(b[i]), "+v"(b[(i + 1) % CHAINS]) : "s"(c)); // 2 VOP2 + 1 VOP3
        }
    }
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ b[i] ^ (u32)w[i] ^ (u32)(w[i] >> 32);
    if (r == 0x12345678u) out[t] = r; // (keeps the chains alive)
}

template <int OP>
static void run(const char *name, int n_instr_per_slot, u32 *d_out) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * (getenv("WPS") ? atoi(getenv("WPS")) : 8); // workgroups of 4 waves per CU = waves per SIMD (WPS)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u + r, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * N_ITER * CHAINS * n_instr_per_slot * reps; // wave-level instructions
    const double per_simd_per_s = wave_instr / (cus * 4) / (ms * 1e-3);
    printf("%-28s %8.3f ms  %7.2f G wave-instr/s/SIMD  -> %5.2f cycles per wave-instruction at 2.4 GHz  (lane-ops %.1f T/s, CUs %d, clock %d MHz)\n", name, ms / reps,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e12, cus, p.clockRate / 1000);
}

int main() {
    u32 *d_out;
    hipMalloc(&d_out, 256u * 8 * 256 * 4 * 4);
    run<8>("v_xor_b32", 1, d_out);
    run<5>("v_add3_u32", 1, d_out);
    run<6>("v_alignbit_b32", 1, d_out);
    run<12>("v_xad_u32", 1, d_out);
    run<7>("v_lshl_add_u64", 1, d_out);
    run<11>("v_lshrrev_b64", 1, d_out);
    run<0>("v_mul_lo_u32", 1, d_out);
    run<1>("v_mul_hi_u32", 1, d_out);
    run<2>("v_mad_u64_u32 (acc)", 1, d_out);
    run<10>("v_mad_u64_u32 (+0)", 1, d_out);
    run<3>("v_mul_u32_u24", 1, d_out);
    run<9>("v_mul_hi_u32_u24", 1, d_out);
    run<4>("v_mad_u32_u24", 1, d_out);
    run<13>("v_mul_lo_u32 + v_xor", 2, d_out);
    run<14>("v_dot4_u32_u8", 1, d_out);
    run<15>("v_pk_mul_lo_u16", 1, d_out);
    run<16>("v_pk_mad_u16", 1, d_out);
    run<17>("v_mul_f64", 1, d_out);
    run<18>("v_fma_f64", 1, d_out);
    run<19>("v_xor_b32_e64 (VOP3)", 1, d_out);
    run<20>("v_add_u32_e32", 1, d_out);
    run<21>("v_lshrrev_b32_e32", 1, d_out);
    run<22>("v_and_or_b32", 1, d_out);
    run<23>("v_add_co + v_addc (VOP2 x2)", 2, d_out);
    run<24>("v_xor_b32_e32 + literal", 1, d_out);
    run<26>("2 x v_xor_e32 + v_mul_lo", 3, d_out);
    return 0;
}
