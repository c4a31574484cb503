// Micro-benchmark, part 2: which VALU opcodes / operand forms issue at the full rate (one wave64 instruction per ~2.5 clocks once a
// SIMD holds >= 2 waves) and which at half or quarter rate, on gfx950.  Same method as valu_rate.hip (residency pinned by LDS, 64
// instructions per loop trip on 8 independent registers), one row per opcode form, at 1 / 2 / 4 waves per SIMD:
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate2.hip -o /tmp/valu_rate2 && /tmp/valu_rate2 > profiles/r03_valu_rate2.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// N(name, text): %0 = the 32-bit register of the chain, %1 / %2 two other VGPRs, s24 a uniform SGPR, s[26:27] a lane mask
// W(name, text): the same on a 64-bit register pair
#define OPS(N, W) \
    N(v_add_f32,            "v_add_f32 %0, %0, %1") \
    N(v_sub_f32,            "v_sub_f32 %0, %0, %1") \
    N(v_mul_f32,            "v_mul_f32 %0, %0, %1") \
    N(v_mul_f32_sgpr,       "v_mul_f32 %0, s24, %0") \
    N(v_mul_f32_inline_const, "v_mul_f32 %0, 0.5, %0") \
    N(v_mul_f32_literal,    "v_mul_f32 %0, 0x3f7fff00, %0") \
    N(v_fmac_f32,           "v_fmac_f32 %0, %1, %2") \
    N(v_fma_f32,            "v_fma_f32 %0, %0, %1, %2") \
    N(v_fma_f32_neg_mod,    "v_fma_f32 %0, -%0, %1, %2") \
    N(v_fma_f32_sgpr,       "v_fma_f32 %0, %0, s24, %2") \
    N(v_fma_f32_inline_const, "v_fma_f32 %0, %0, %1, 1.0") \
    N(v_fmaak_f32,          "v_fmaak_f32 %0, %0, %1, 0x3f7fff00") \
    N(v_max_f32,            "v_max_f32 %0, %0, %1") \
    N(v_min_f32,            "v_min_f32 %0, %0, %1") \
    N(v_med3_f32,           "v_med3_f32 %0, %0, %1, %2") \
    N(v_mov_b32,            "v_mov_b32 %0, %1") \
    N(v_and_b32,            "v_and_b32 %0, %0, %1") \
    N(v_or_b32,             "v_or_b32 %0, %0, %1") \
    N(v_xor_b32,            "v_xor_b32 %0, %0, %1") \
    N(v_lshlrev_b32,        "v_lshlrev_b32 %0, 1, %0") \
    N(v_lshrrev_b32,        "v_lshrrev_b32 %0, 1, %0") \
    N(v_add_u32,            "v_add_u32 %0, %0, %1") \
    N(v_sub_u32,            "v_sub_u32 %0, %0, %1") \
    N(v_add3_u32,           "v_add3_u32 %0, %0, %1, %2") \
    N(v_lshl_add_u32,       "v_lshl_add_u32 %0, %0, 2, %1") \
    N(v_and_or_b32,         "v_and_or_b32 %0, %0, %1, %2") \
    N(v_bfe_u32,            "v_bfe_u32 %0, %0, 3, 8") \
    N(v_mul_lo_u32,         "v_mul_lo_u32 %0, %0, %1") \
    N(v_mul_u32_u24,        "v_mul_u32_u24 %0, %0, %1") \
    N(v_mad_u32_u24,        "v_mad_u32_u24 %0, %0, %1, %2") \
    N(v_cvt_f32_u32,        "v_cvt_f32_u32 %0, %0") \
    N(v_cvt_u32_f32,        "v_cvt_u32_f32 %0, %0") \
    N(v_floor_f32,          "v_floor_f32 %0, %0") \
    N(v_rndne_f32,          "v_rndne_f32 %0, %0") \
    N(v_ldexp_f32,          "v_ldexp_f32 %0, %0, %1") \
    N(v_cmp_gt_f32_vcc,     "v_cmp_gt_f32 vcc, %0, %1") \
    N(v_cmp_gt_f32_sgpr,    "v_cmp_gt_f32 s[28:29], %0, %1") \
    N(v_cndmask_sgpr_mask,  "v_cndmask_b32_e64 %0, %0, %1, s[26:27]") \
    N(v_mov_dpp_quad_perm,  "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    N(v_mov_dpp_row_ror,    "v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf") \
    N(v_mov_dpp_row_bcast15, "v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf") \
    N(v_add_f32_dpp_quad_perm, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    N(v_readfirstlane_b32,  "v_readfirstlane_b32 s28, %0") \
    N(v_exp_f32,            "v_exp_f32 %0, %0") \
    N(v_rsq_f32,            "v_rsq_f32 %0, %0") \
    N(v_sin_f32,            "v_sin_f32 %0, %0") \
    W(v_add_f64,            "v_add_f64 %0, %0, %0") \
    W(v_fma_f64,            "v_fma_f64 %0, %0, %0, %0") \
    W(v_pk_fma_f32,         "v_pk_fma_f32 %0, %0, %0, %0") \
    W(v_pk_mul_f32,         "v_pk_mul_f32 %0, %0, %0") \
    W(v_pk_add_f32,         "v_pk_add_f32 %0, %0, %0") \
    W(v_pk_mov_b32,         "v_pk_mov_b32 %0, %0, %0") \
    N(s_nop,                "s_nop 0")

#define N(name, str) name,
#define W(name, str) name,
enum Op { OPS(N, W) N_OPS };
#undef N
#undef W
#define N(name, str) #name,
#define W(name, str) #name,
static const char* op_name[] = { OPS(N, W) };
#undef N
#undef W
#define N(name, str) str,
#define W(name, str) str,
static const char* op_text[] = { OPS(N, W) };
#undef N
#undef W

template <int OP> struct Body;
#define R8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)
#define N(name, str) template <> struct Body<name> { static __device__ __forceinline__ void run(float (&a)[8], double (&d)[8], float c0, float c1) { \
    _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(str : "+v"(a[i]) : "v"(c0), "v"(c1) : "s28", "s29", "vcc"); } };
#define W(name, str) template <> struct Body<name> { static __device__ __forceinline__ void run(float (&a)[8], double (&d)[8], float c0, float c1) { \
    _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(str : "+v"(d[i]) : "v"(c0), "v"(c1) : "s28", "s29", "vcc"); } };
OPS(N, W)
#undef N
#undef W

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(int iters, float* sink, unsigned long long* stamps) {
    extern __shared__ float4 lds[];
    float a[8]; double d[8];
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + 0.001f * (float)(threadIdx.x + i); d[i] = 1.0 + 0.001 * (double)(threadIdx.x + i); }
    const float c0 = 0.999999f, c1 = 1e-7f;
    asm volatile("s_mov_b32 s24, 0x3f7ffff0\n s_mov_b64 s[26:27], 0x5555\n" ::: "s24", "s26", "s27");
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1);
        Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1); Body<OP>::run(a, d, c0, c1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {      // chip-wide span of the loop in both clocks: min start, max end over all waves
        atomicMin(stamps + 0, t0); atomicMax(stamps + 1, t1); atomicMin(stamps + 2, r0); atomicMax(stamps + 3, r1);
    }
}

typedef void (*kern_t)(int, float*, unsigned long long*);
template <int... I> static std::vector<kern_t> table(std::integer_sequence<int, I...>) { return {k_rate<I>...}; }

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    std::vector<kern_t> ks = table(std::make_integer_sequence<int, N_OPS>{});
    float* d_sink; CK(hipMalloc(&d_sink, 4));
    unsigned long long* d_st; CK(hipMalloc(&d_st, 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"instructions_per_wave\": %d,\n", prop.gcnArchName, CUS, iters * 64);
    printf(" \"note\": \"cyc = wall time per wave-instruction per SIMD x 2.4 GHz (nominal clock); residency pinned by LDS (one workgroup of 256 x w threads per CU)\",\n \"results\": [\n");
    for (int op = 0; op < N_OPS; op++) {
        printf("  {\"op\": \"%s\", \"text\": \"%s\"", op_name[op], op_text[op]);
        for (int wps : {1, 2, 4}) {
            const size_t lds = 96 * 1024;
            CK(hipFuncSetAttribute((const void*)ks[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            float ms = 0;
            unsigned long long st[4];
            for (int rep = 0; rep < 2; rep++) {
                const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
                CK(hipMemcpy(d_st, init, 32, hipMemcpyHostToDevice));
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(ks[op], dim3(CUS), dim3(256 * wps), lds, 0, iters, d_sink, d_st);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(st, d_st, 32, hipMemcpyDeviceToHost));
            }
            const double inst = (double)iters * 64;
            const double nw = (double)CUS * 4 * wps;
            const double ticks = (double)(st[1] - st[0]), real = (double)(st[3] - st[2]);      // s_memtime | s_memrealtime (100 MHz) spans
            printf(", \"w%d\": {\"ginst_per_s\": %.1f, \"memtime_ticks_per_inst_per_simd\": %.3f, \"memtime_MHz\": %.0f, \"wall_ms\": %.3f}", wps,
                   inst * nw / ((double)ms * 1e6), ticks / inst / wps, ticks / real * 100.0, ms);
        }
        printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    printf(" ]\n}\n");
    return 0;
}
