// Issue-rate microbenchmark for the integer VALU / DPP / LDS-permute instructions the np2 kernels are made of
// (gfx950).  Every test runs the same instruction 64 x per loop iteration over 8 independent register chains, on enough
// waves to fill every SIMD, and reports SIMD-cycles per wave-instruction = elapsed cycles x SIMDs / wave-instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/bin/ubench_valu ; tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr int ITER = 2048;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP64(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL32(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                 \
        uint32_t r[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                \
        for (int i = 0; i < 8; ++i) r[i] = seed + i + threadIdx.x;                              \
        for (int it = 0; it < ITER; ++it) {                                                     \
            REP64(ASM)                                                                          \
        }                                                                                       \
        uint32_t s = 0;                                                                         \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                  \
        if (s == 0x12345u) out[threadIdx.x] = s + b + c;                                        \
    }

#define A_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_XOR3(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r[i]));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(b));
#define A_AND_OR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(b), "v"(c));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r[i]));
#define A_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_DPPADD(i) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
#define A_DPPBC(i) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:31 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(b));
#define A_PERMLANE(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(b));
#define A_READLANE(i) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(r[i])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "s"(s_)); }
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : );
#define A_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(b) : "vcc");

KERNEL32(k_add, A_ADD)
KERNEL32(k_and, A_AND)
KERNEL32(k_xor, A_XOR3)
KERNEL32(k_alignbit, A_ALIGN)
KERNEL32(k_bcnt, A_BCNT)
KERNEL32(k_ffbl, A_FFBL)
KERNEL32(k_perm, A_PERM)
KERNEL32(k_mullo, A_MULLO)
KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_and_or, A_AND_OR)
KERNEL32(k_bfe, A_BFE)
KERNEL32(k_dpp_mov, A_DPP)
KERNEL32(k_dpp_add, A_DPPADD)
KERNEL32(k_dpp_bcast, A_DPPBC)
KERNEL32(k_bpermute, A_BPERM)
KERNEL32(k_permlane32_swap, A_PERMLANE)
KERNEL32(k_readlane_add, A_READLANE)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_cmp, A_CMP)

// 64-bit forms
#define KERNEL64(NAME, ASM)                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) {                 \
        uint64_t r[8], b = seed + threadIdx.x;                                                  \
        uint32_t c = (seed & 31) + 1;                                                           \
        for (int i = 0; i < 8; ++i) r[i] = seed + i + threadIdx.x;                              \
        for (int it = 0; it < ITER; ++it) {                                                     \
            REP64(ASM)                                                                          \
        }                                                                                       \
        uint64_t s = 0;                                                                         \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                  \
        if (s == 0x12345u) out[threadIdx.x] = (uint32_t)(s + b + c);                            \
    }
#define B_SHL(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define B_SHR(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define B_ADD(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(b));
#define B_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(*(uint32_t *)&r[i]) : "v"((uint32_t)b));
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(r[i]) : "v"(c) : "vcc");
KERNEL64(k_lshl_b64, B_SHL)
KERNEL64(k_lshr_b64, B_SHR)
KERNEL64(k_add_u64, B_ADD)
KERNEL64(k_pk_add_u16, B_PKADD)
KERNEL64(k_mad_u64_u32, B_MAD64)

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Test { const char *name; kern_t k; int per_iter; };

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("device %s: %d CUs, %.0f MHz\n", p.gcnArchName, cus, mhz);
    uint32_t *out;
    CHK(hipMalloc(&out, 4096));
    Test tests[] = {
        {"v_add_u32", k_add, 64}, {"v_and_b32", k_and, 64}, {"v_xor_b32", k_xor, 64}, {"v_alignbit_b32", k_alignbit, 64},
        {"v_bcnt_u32_b32", k_bcnt, 64}, {"v_ffbl_b32", k_ffbl, 64}, {"v_perm_b32", k_perm, 64}, {"v_mul_lo_u32", k_mullo, 64},
        {"v_lshl_add_u32", k_lshladd, 64}, {"v_and_or_b32", k_and_or, 64}, {"v_bfe_u32", k_bfe, 64},
        {"v_mov_b32_dpp row_shr", k_dpp_mov, 64}, {"v_add_u32_dpp row_shr", k_dpp_add, 64}, {"v_mov_b32_dpp row_bcast31", k_dpp_bcast, 64},
        {"ds_bpermute_b32 (+wait)", k_bpermute, 64}, {"v_permlane32_swap", k_permlane32_swap, 64},
        {"v_readlane + v_add(s)", k_readlane_add, 128}, {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp_lt_u32", k_cmp, 64},
        {"v_lshlrev_b64", k_lshl_b64, 64}, {"v_lshrrev_b64", k_lshr_b64, 64}, {"v_lshl_add_u64", k_add_u64, 64},
        {"v_pk_add_u16", k_pk_add_u16, 64}, {"v_mad_u64_u32", k_mad_u64_u32, 64},
    };
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    for (int wps : {1, 2, 4, 8}) { // waves per SIMD
        printf("-- %d wave(s) per SIMD --\n", wps);
        for (const Test &t : tests) {
            const int blocks = cus * wps; // 256-thread block = 4 waves = one per SIMD
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 7u);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 7u);
            CHK(hipEventRecord(b, 0));
            CHK(hipEventSynchronize(b));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, a, b));
            const double winstr = (double)blocks * 4 * ITER * t.per_iter;
            const double cyc = ms * 1e-3 * mhz * 1e6;
            printf("%-28s %8.3f ms  %6.2f SIMD-cycles per wave-instruction\n", t.name, ms, cyc * cus * 4 / winstr);
        }
    }
    return 0;
}
