// Micro-benchmark: issue rate of the instruction classes the blend / preprocess kernels are made of, on gfx950 (MI355X).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > profiles/r03_valu_rate.json
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -- /tmp/valu_rate pmc
//
// Why: bench.py priced every launch against a VALU-issue roof of 256 CU x 4 SIMD x 2.4 GHz / 4 clocks per wave64 instruction
// (614 G wave-instructions/s).  MI355X_MICROARCH.md says the SIMDs are 32 lanes wide and a wave64 v_fma_f32 issues in 2 clocks.
// This measures it: for each instruction class and 1 / 2 / 4 / 8 resident waves per SIMD, the cycles one wave needs per
// instruction (s_memtime around a long unrolled loop, so independent of the clock the chip happens to run at) and the wall-clock
// rate of the whole chip (hip events).  Residency is pinned by LDS: a workgroup asks for so much dynamic LDS that exactly one
// (or two) fit on a CU, and the grid is one (two) workgroup per CU, so every SIMD holds exactly the stated number of waves.
//
// Output: one JSON object on stdout.  "cyc_per_inst_per_simd" = wave cycles per instruction / waves per SIMD = the issue interval
// of the SIMD when the waves are symmetric; "ginst_per_s" = wave-instructions per second of the chip, wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Op { FMA_INDEP, FMA_DEP, MUL_INDEP, PK_FMA, PK_MUL, PK_ADD, EXP, RCP, SQRT, LOG, MOV_DPP, ADD_DPP, CNDMASK, CMP, PERMLANE32_SWAP, BITOP3,
          CVT_I32, FMA_EXP_MIX, DS_READ_B128, DS_READ_B32, READLANE, BALLOT_CMP, NOPS,
          CNDMASK_SGPR, CMP_CNDMASK, MAX_F32, FMAC, MOV, ADD_F32, FMA_SGPR, MED3, CMP_SGPR_CNDMASK };
static const char* op_name[] = {"v_fma_f32 (8 independent chains)", "v_fma_f32 (one dependent chain)", "v_mul_f32 (8 chains)", "v_pk_fma_f32 (8 chains)",
                                "v_pk_mul_f32 (8 chains)", "v_pk_add_f32 (8 chains)", "v_exp_f32 (8 chains)", "v_rcp_f32 (8 chains)",
                                "v_sqrt_f32 (8 chains)", "v_log_f32 (8 chains)", "v_mov_b32 dpp row_shr:1 (8 chains)", "v_add_f32 dpp row_shr:1 (8 chains)",
                                "v_cndmask_b32 (8 chains)", "v_cmp_gt_f32 -> vcc (8 sources)", "v_permlane32_swap (4 pairs)",
                                "v_bitop3_b32 (8 chains)", "v_cvt_i32_f32 (8 chains)", "3 v_fma_f32 : 1 v_exp_f32 (8 chains)", "ds_read_b128 (8 targets)",
                                "ds_read_b32 (8 targets)", "v_readlane_b32 -> sgpr (8 sources)", "v_cmp_gt_f32 -> sgpr pair + s_and (ballot use)", "s_nop 0",
                                "v_cndmask_b32_e64 mask in an sgpr pair (8 chains)", "v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (select idiom, 4 pairs)",
                                "v_max_f32 (8 chains)", "v_fmac_f32 (8 chains)", "v_mov_b32 (8 chains)", "v_add_f32 (8 chains)",
                                "v_fma_f32 with an sgpr operand (8 chains)", "v_med3_f32 (8 chains)",
                                "v_cmp_gt_f32 sgpr pair + v_cndmask_b32_e64 (select idiom, 4 pairs)"};
constexpr int N_OPS = sizeof(op_name) / sizeof(op_name[0]);
#define UNROLL 8
static_assert(UNROLL == 8, "");
//       // groups of 8 instructions per loop body -> 64 instructions + loop overhead (s_sub, s_cmp, s_cbranch)

// 8 instructions of the class on 8 different registers (or one, for the dependent chain)
#define G8(fmt) asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c0), "v"(c1) : "s20", "s21", "s22", "s23", "s24", "s25", "vcc", "scc")
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_FMA_D(i) "v_fma_f32 %0, %0, %8, %9\n"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define I_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CMP(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define I_BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define I_CVT(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define I_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define I_CND_S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[24:25]\n"
#define I_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define I_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", s24, %9\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_BALLOT(i) "v_cmp_gt_f32 s[20:21], %" #i ", %8\n s_and_b64 s[22:23], s[20:21], exec\n"

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(int iters, unsigned long long* cycles, float* sink) {
    extern __shared__ float4 lds[];
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
    const float c0 = 0.999999f, c1 = 1e-7f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
    for (int i = 0; i < 8; i++) p[i] = v2f{a[i], a[i] + 0.5f};
    const v2f pc0 = {c0, c0}, pc1 = {c1, c1};
    if (OP == DS_READ_B128 || OP == DS_READ_B32) {
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(i, 1, 2, 3);
        __syncthreads();
    }
    const unsigned addr = (threadIdx.x & 63) * 16u;          // conflict-free b128 pattern: 64 lanes x 16 B contiguous
    float4 r[8];
    __syncthreads();
    asm volatile("s_mov_b64 s[24:25], 0x5555\n" ::: "s24", "s25");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it++) {
        auto body = [&]() __attribute__((always_inline)) {
            if (OP == FMA_INDEP) G8(I_FMA);
            else if (OP == FMA_DEP) G8(I_FMA_D);
            else if (OP == MUL_INDEP) G8(I_MUL);
            else if (OP == EXP) G8(I_EXP);
            else if (OP == RCP) G8(I_RCP);
            else if (OP == SQRT) G8(I_SQRT);
            else if (OP == LOG) G8(I_LOG);
            else if (OP == MOV_DPP) G8(I_MOVDPP);
            else if (OP == ADD_DPP) G8(I_ADDDPP);
            else if (OP == CNDMASK) G8(I_CND);
            else if (OP == CMP) G8(I_CMP);
            else if (OP == BITOP3) G8(I_BITOP);
            else if (OP == CVT_I32) G8(I_CVT);
            else if (OP == READLANE) G8(I_READLANE);
            else if (OP == BALLOT_CMP) G8(I_BALLOT);
            else if (OP == CNDMASK_SGPR) G8(I_CND_S);
            else if (OP == MAX_F32) G8(I_MAX);
            else if (OP == FMAC) G8(I_FMAC);
            else if (OP == MOV) G8(I_MOV);
            else if (OP == ADD_F32) G8(I_ADD);
            else if (OP == FMA_SGPR) G8(I_FMA_S);
            else if (OP == MED3) G8(I_MED3);
            else if (OP == CMP_CNDMASK) {
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                             "v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c0), "v"(c1) : "vcc");
            } else if (OP == CMP_SGPR_CNDMASK) {
                asm volatile("v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %8\n v_cndmask_b32_e64 %0, %0, %9, s[20:21]\n v_cndmask_b32_e64 %1, %1, %9, s[22:23]\n"
                             "v_cmp_gt_f32 s[20:21], %2, %8\n v_cmp_gt_f32 s[22:23], %3, %8\n v_cndmask_b32_e64 %2, %2, %9, s[20:21]\n v_cndmask_b32_e64 %3, %3, %9, s[22:23]\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c0), "v"(c1) : "s20", "s21", "s22", "s23");
            }
            else if (OP == FMA_EXP_MIX) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c0), "v"(c1));
            } else if (OP == PK_FMA || OP == PK_MUL || OP == PK_ADD) {
#define PK8(ins, tail) asm volatile(ins " %0, %0, %8" tail "\n" ins " %1, %1, %8" tail "\n" ins " %2, %2, %8" tail "\n" ins " %3, %3, %8" tail "\n" \
                                    ins " %4, %4, %8" tail "\n" ins " %5, %5, %8" tail "\n" ins " %6, %6, %8" tail "\n" ins " %7, %7, %8" tail "\n" \
                                    : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pc0), "v"(pc1))
                if (OP == PK_FMA) PK8("v_pk_fma_f32", ", %9");
                else if (OP == PK_MUL) PK8("v_pk_mul_f32", "");
                else PK8("v_pk_add_f32", "");
            } else if (OP == PERMLANE32_SWAP) {
#pragma unroll
                for (int rep = 0; rep < 2; rep++)
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 1]), false, false);
                        a[i] = __uint_as_float(sw[0]); a[i + 1] = __uint_as_float(sw[1]);
                        asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]));
                    }
            } else if (OP == DS_READ_B128) {
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                             "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(addr) : "memory");
            } else if (OP == DS_READ_B32) {
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                             "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]) : "v"(addr / 4u) : "memory");
            } else if (OP == NOPS) {
                asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
            }
        };
        body(); body(); body(); body(); body(); body(); body(); body();       // UNROLL = 8, written out: the unroller leaves loops around inline asm alone
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    if (OP == DS_READ_B128) for (int i = 0; i < 8; i++) s += r[i].x + r[i].w;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(int, unsigned long long*, float*);
template <int OP> static kern_t get() { return k_rate<OP>; }
template <int... I> static std::vector<kern_t> table(std::integer_sequence<int, I...>) { return {get<I>()...}; }

int main(int argc, char** argv) {
    const bool pmc_mode = argc > 1 && !strcmp(argv[1], "pmc");       // one short launch per (op, residency): for the counter pass
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    // s_memtime runs at a fixed 100 MHz reference on gfx9 (not the shader clock): calibrate it against hip events on the spot
    std::vector<kern_t> ks = table(std::make_integer_sequence<int, N_OPS>{});
    unsigned long long* d_cyc; float* d_sink;
    CK(hipMalloc(&d_cyc, 8 * 8192)); CK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(8192);
    const int iters = pmc_mode ? 2000 : 20000;
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"instructions_per_wave\": %d,\n", prop.gcnArchName, CUS, prop.clockRate, iters * UNROLL * 8);
    printf(" \"note\": \"wave-level issue rates; cyc = s_memtime ticks scaled to shader cycles by the measured tick/wall ratio and the nominal 2.4 GHz; "
           "ginst_per_s is wall-clock and includes DVFS; residency pinned by LDS\",\n \"results\": [\n");
    bool first = true;
    for (int op = 0; op < N_OPS; op++) {
        for (int wps : {1, 2, 4, 8}) {
            int threads, blocks_per_cu; size_t lds;
            if (wps <= 4) { threads = 256 * wps; blocks_per_cu = 1; lds = 96 * 1024; }      // one workgroup per CU (2 x 96 KB > 160 KB)
            else { threads = 1024; blocks_per_cu = 2; lds = 64 * 1024; }                    // two per CU (3 x 64 KB > 160 KB)
            CK(hipFuncSetAttribute((const void*)ks[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = CUS * blocks_per_cu;
            const int nw = grid * (threads / 64);
            for (int rep = 0; rep < (pmc_mode ? 1 : 2); rep++) {                             // first rep = warm-up / clock ramp
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(ks[op], dim3(grid), dim3(threads), lds, 0, iters, d_cyc, d_sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_cyc, 8 * (size_t)nw, hipMemcpyDeviceToHost));
            double sum = 0, mx = 0;
            for (int i = 0; i < nw; i++) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
            const double inst = (double)iters * UNROLL * 8;                                  // per wave
            const double ticks_per_inst = sum / nw / inst;                                   // s_memtime ticks (100 MHz)
            const double wall_ns_per_inst_per_wave = (double)ms * 1e6 / inst;
            const double ginst = inst * nw / ((double)ms * 1e6);
            // shader cycles per instruction per wave at the nominal clock, from wall time (the kernel is one long loop: launch overhead < 0.1 %)
            const double cyc_nominal = wall_ns_per_inst_per_wave * 2.4;
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ginst_per_s\": %.1f, \"cyc_per_inst_per_wave_at_2p4GHz\": %.3f, "
                   "\"cyc_per_inst_per_simd_at_2p4GHz\": %.3f, \"memtime_ticks_per_inst\": %.5f, \"slowest_wave_over_mean\": %.3f}",
                   first ? "" : ",\n", op_name[op], wps, ms, ginst, cyc_nominal, cyc_nominal / wps, ticks_per_inst, mx / (sum / nw));
            first = false;
        }
    }
    printf("\n ]\n}\n");
    return 0;
}
