// valu_issue.hip -- per-opcode VALU issue cost on gfx950 in SHADER CYCLES, and the clock the chip sustains meanwhile.
//
// Why a second tool next to valu_peak.hip: valu_peak.hip divides wall time by instruction count, so a DVFS clock drop
// (the chip clocks to its power budget: MI355X_MICROARCH.md "DVFS give-back") and a real multi-cycle issue cost look
// the same.  Here every wave brackets its instruction stream with s_memtime (shader-clock ticks) and s_memrealtime
// (constant 100 MHz), so each test reports
//     cycles per wave-instruction per SIMD  = (last s_memtime - first s_memtime on that SIMD) / instructions issued there
//     sustained clock                        = d(memtime) / d(memrealtime) * 100 MHz
// separately.  Waves are grouped by the SIMD they really ran on (HW_REG_HW_ID + HW_REG_XCC_ID).
//
// Each test body is one inline-asm block of 64 VALU instructions on 8 independent accumulators (dependent distance 8),
// so hipcc schedules nothing in between; the surrounding loop adds 3 SALU instructions per 64 VALU.
//
// Build:  hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/valu_issue
// Run:    tools/valu_issue > profiles/round2/valu_issue_mi355x.jsonl
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct WaveRec { uint64_t t0, t1, r0, r1; uint32_t hwid, xcc; };

typedef float f2 __attribute__((ext_vector_type(2)));

// X(i) = accumulator i ("%0".."%7"); inputs: %8 = a (VGPR), %9 = b (VGPR), %10 = s (SGPR), %11 = mask (SGPR pair)
#define A0 "%0"
#define A1 "%1"
#define A2 "%2"
#define A3 "%3"
#define A4 "%4"
#define A5 "%5"
#define A6 "%6"
#define A7 "%7"
#define ALL8(I) I(A0) I(A1) I(A2) I(A3) I(A4) I(A5) I(A6) I(A7)

#define I_FMA_VVV(X) "v_fma_f32 " X ", " X ", %8, %9\n"
#define I_FMA_ACC(X) "v_fma_f32 " X ", %8, %9, " X "\n"
#define I_FMAC(X) "v_fmac_f32 " X ", %8, %9\n"
#define I_FMA_SGPR(X) "v_fma_f32 " X ", " X ", %10, %9\n"
#define I_FMA_SS(X) "v_fma_f32 " X ", " X ", %10, %10\n"
#define I_FMA_CONST(X) "v_fma_f32 " X ", " X ", 1.0, 0.5\n"
#define I_FMA_CLAMP(X) "v_fma_f32 " X ", " X ", %8, %9 clamp\n"
#define I_FMAAK(X) "v_fmaak_f32 " X ", " X ", %8, 0x3f8ccccd\n"
#define I_MUL(X) "v_mul_f32 " X ", " X ", %8\n"
#define I_MUL_E64(X) "v_mul_f32_e64 " X ", " X ", %8\n"
#define I_ADD(X) "v_add_f32 " X ", " X ", %9\n"
#define I_SUB(X) "v_sub_f32 " X ", " X ", %9\n"
#define I_MAX(X) "v_max_f32 " X ", " X ", %9\n"
#define I_MIN(X) "v_min_f32 " X ", " X ", %9\n"
#define I_MED3(X) "v_med3_f32 " X ", " X ", 0, 1.0\n"
#define I_MAX3(X) "v_max3_f32 " X ", " X ", %8, %9\n"
#define I_CVT_UB0(X) "v_cvt_f32_ubyte0 " X ", " X "\n"
#define I_CVT_UB2(X) "v_cvt_f32_ubyte2 " X ", " X "\n"
#define I_CVT_I32(X) "v_cvt_i32_f32 " X ", " X "\n"
#define I_CVT_F32_I32(X) "v_cvt_f32_i32 " X ", " X "\n"
#define I_CVT_F32_U32(X) "v_cvt_f32_u32 " X ", " X "\n"
#define I_FLOOR(X) "v_floor_f32 " X ", " X "\n"
#define I_FRACT(X) "v_fract_f32 " X ", " X "\n"
#define I_CNDMASK_VCC(X) "v_cndmask_b32 " X ", " X ", %8, vcc\n"
#define I_CNDMASK_S(X) "v_cndmask_b32_e64 " X ", " X ", %8, %11\n"
#define I_CMP(X) "v_cmp_lt_f32 vcc, " X ", %8\n"
#define I_CMP_CND(X) "v_cmp_lt_f32 vcc, " X ", %8\nv_cndmask_b32 " X ", " X ", %9, vcc\n"
#define I_MOV(X) "v_mov_b32 " X ", %8\n"
#define I_AND(X) "v_and_b32 " X ", " X ", %8\n"
#define I_LSHL(X) "v_lshlrev_b32 " X ", 2, " X "\n"
#define I_ADD_U32(X) "v_add_u32 " X ", " X ", %8\n"
#define I_ADD3_U32(X) "v_add3_u32 " X ", " X ", %8, %9\n"
#define I_LSHL_ADD(X) "v_lshl_add_u32 " X ", " X ", 2, %8\n"
#define I_MAD_U24(X) "v_mad_u32_u24 " X ", " X ", %8, %9\n"
#define I_MUL_LO(X) "v_mul_lo_u32 " X ", " X ", %8\n"
#define I_BFE(X) "v_bfe_u32 " X ", " X ", 8, 8\n"
#define I_EXP(X) "v_exp_f32 " X ", " X "\n"
#define I_LOG(X) "v_log_f32 " X ", " X "\n"
#define I_SQRT(X) "v_sqrt_f32 " X ", " X "\n"
#define I_RSQ(X) "v_rsq_f32 " X ", " X "\n"
#define I_RCP(X) "v_rcp_f32 " X ", " X "\n"
#define I_CUBEID(X) "v_cubeid_f32 " X ", " X ", %8, %9\n"
#define I_CUBESC(X) "v_cubesc_f32 " X ", " X ", %8, %9\n"
#define I_CUBEMA(X) "v_cubema_f32 " X ", " X ", %8, %9\n"
#define I_LDEXP(X) "v_ldexp_f32 " X ", " X ", 1\n"


#define I_MUL_SGPR(X) "v_mul_f32 " X ", %10, " X "\n"
#define I_ADD_SGPR(X) "v_add_f32 " X ", %10, " X "\n"
#define I_FMAC_SGPR(X) "v_fmac_f32 " X ", %10, %9\n"
#define I_MUL_LIT(X) "v_mul_f32 " X ", 0x3f800001, " X "\n"
#define I_FMA_NEG(X) "v_fma_f32 " X ", " X ", -%8, %9\n"
#define I_FMA_ABS(X) "v_fma_f32 " X ", |" X "|, %8, %9\n"
#define I_MUL_OMOD(X) "v_mul_f32_e64 " X ", " X ", %8 mul:2\n"
#define I_OR(X) "v_or_b32 " X ", " X ", %8\n"
#define I_XOR(X) "v_xor_b32 " X ", " X ", %8\n"
#define I_LSHR(X) "v_lshrrev_b32 " X ", 8, " X "\n"
#define I_ASHR(X) "v_ashrrev_i32 " X ", 1, " X "\n"
#define I_SUB_U32(X) "v_sub_u32 " X ", " X ", %8\n"
#define I_AND_OR(X) "v_and_or_b32 " X ", " X ", %8, %9\n"
#define I_OR3(X) "v_or3_b32 " X ", " X ", %8, %9\n"
#define I_PERM(X) "v_perm_b32 " X ", " X ", %8, %9\n"
#define I_RNDNE(X) "v_rndne_f32 " X ", " X "\n"
#define I_TRUNC(X) "v_trunc_f32 " X ", " X "\n"
#define I_MUL_U24(X) "v_mul_u32_u24 " X ", " X ", %8\n"
#define I_CVT_U32(X) "v_cvt_u32_f32 " X ", " X "\n"
#define I_MOV_DPP(X) "v_mov_b32_dpp " X ", " X " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_ADD_DPP(X) "v_add_f32_dpp " X ", " X ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_SUB_DPP(X) "v_sub_f32_dpp " X ", " X ", " X " quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n"
#define I_CVT_UB1_SDWA(X) "v_cvt_f32_ubyte1 " X ", " X "\n"
#define I_MIN3(X) "v_min3_f32 " X ", " X ", %8, %9\n"
#define I_SQRT_F16(X) "v_sqrt_f16 " X ", " X "\n"
#define I_EXP_F16(X) "v_exp_f16 " X ", " X "\n"
#define I_MAD_U64(X) "v_mul_hi_u32 " X ", " X ", %8\n"

struct Args { float a, b, s; uint64_t mask; };

#define BODY_INPUTS : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                    : "v"(g.a), "v"(g.b), "s"(g.s), "s"(g.mask) : "vcc"

// a test = a struct with body(x[8], Args) issuing N VALU instructions
#define SINGLE(NAME, INS)                                                      \
    struct NAME {                                                              \
        static constexpr int N = 64;                                           \
        static const char *name() { return #INS; }                             \
        __device__ static __forceinline__ void body(float *x, const Args &g) { \
            asm volatile(".rept 8\n" ALL8(INS) ".endr\n" BODY_INPUTS);         \
        }                                                                      \
    };

SINGLE(T_fma_vvv, I_FMA_VVV) SINGLE(T_fma_acc, I_FMA_ACC) SINGLE(T_fmac, I_FMAC) SINGLE(T_fma_sgpr, I_FMA_SGPR)
SINGLE(T_fma_ss, I_FMA_SS) SINGLE(T_fma_const, I_FMA_CONST) SINGLE(T_fma_clamp, I_FMA_CLAMP) SINGLE(T_fmaak, I_FMAAK)
SINGLE(T_mul, I_MUL) SINGLE(T_mul_e64, I_MUL_E64) SINGLE(T_add, I_ADD) SINGLE(T_sub, I_SUB) SINGLE(T_max, I_MAX) SINGLE(T_min, I_MIN)
SINGLE(T_med3, I_MED3) SINGLE(T_max3, I_MAX3) SINGLE(T_cvt_ub0, I_CVT_UB0) SINGLE(T_cvt_ub2, I_CVT_UB2) SINGLE(T_cvt_i32, I_CVT_I32)
SINGLE(T_cvt_f32_i32, I_CVT_F32_I32) SINGLE(T_cvt_f32_u32, I_CVT_F32_U32) SINGLE(T_floor, I_FLOOR) SINGLE(T_fract, I_FRACT)
SINGLE(T_cndmask_vcc, I_CNDMASK_VCC) SINGLE(T_cndmask_s, I_CNDMASK_S) SINGLE(T_cmp, I_CMP)
SINGLE(T_mov, I_MOV) SINGLE(T_and, I_AND) SINGLE(T_lshl, I_LSHL) SINGLE(T_add_u32, I_ADD_U32) SINGLE(T_add3_u32, I_ADD3_U32)
SINGLE(T_lshl_add, I_LSHL_ADD) SINGLE(T_mad_u24, I_MAD_U24) SINGLE(T_mul_lo, I_MUL_LO) SINGLE(T_bfe, I_BFE)
SINGLE(T_exp, I_EXP) SINGLE(T_log, I_LOG) SINGLE(T_sqrt, I_SQRT) SINGLE(T_rsq, I_RSQ) SINGLE(T_rcp, I_RCP)
SINGLE(T_cubeid, I_CUBEID) SINGLE(T_cubesc, I_CUBESC) SINGLE(T_cubema, I_CUBEMA) SINGLE(T_ldexp, I_LDEXP)


SINGLE(T_mul_sgpr, I_MUL_SGPR) SINGLE(T_add_sgpr, I_ADD_SGPR) SINGLE(T_fmac_sgpr, I_FMAC_SGPR) SINGLE(T_mul_lit, I_MUL_LIT)
SINGLE(T_fma_neg, I_FMA_NEG) SINGLE(T_fma_abs, I_FMA_ABS) SINGLE(T_mul_omod, I_MUL_OMOD) SINGLE(T_or, I_OR) SINGLE(T_xor, I_XOR)
SINGLE(T_lshr, I_LSHR) SINGLE(T_ashr, I_ASHR) SINGLE(T_sub_u32, I_SUB_U32) SINGLE(T_and_or, I_AND_OR) SINGLE(T_or3, I_OR3) SINGLE(T_perm, I_PERM)
SINGLE(T_rndne, I_RNDNE) SINGLE(T_trunc, I_TRUNC) SINGLE(T_mul_u24, I_MUL_U24) SINGLE(T_cvt_u32, I_CVT_U32) SINGLE(T_mov_dpp, I_MOV_DPP)
SINGLE(T_add_dpp, I_ADD_DPP) SINGLE(T_sub_dpp, I_SUB_DPP) SINGLE(T_min3, I_MIN3) SINGLE(T_sqrt_f16, I_SQRT_F16) SINGLE(T_exp_f16, I_EXP_F16)
SINGLE(T_mul_hi, I_MAD_U64)

// v_cmp + v_cndmask pairs (128 instructions per body)
struct T_cmp_cnd {
    static constexpr int N = 128;
    static const char *name() { return "v_cmp_lt_f32 vcc + v_cndmask_b32 (pair)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept 8\n" ALL8(I_CMP_CND) ".endr\n" BODY_INPUTS); }
};

// dependent chain: one accumulator, 64 back-to-back dependent FMAs (latency, 1 wave per SIMD is the interesting row)
struct T_fma_chain {
    static constexpr int N = 64;
    static const char *name() { return "v_fma_f32 dependent chain (distance 1)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept 64\n" I_FMA_VVV(A0) ".endr\n" BODY_INPUTS); }
};
struct T_exp_chain {
    static constexpr int N = 64;
    static const char *name() { return "v_exp_f32 dependent chain (distance 1)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept 64\n" I_EXP(A0) ".endr\n" BODY_INPUTS); }
};
// distance-2 and distance-4 chains
struct T_fma_dist2 {
    static constexpr int N = 64;
    static const char *name() { return "v_fma_f32 dependent distance 2"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept 32\n" I_FMA_VVV(A0) I_FMA_VVV(A1) ".endr\n" BODY_INPUTS); }
};
struct T_fma_dist4 {
    static constexpr int N = 64;
    static const char *name() { return "v_fma_f32 dependent distance 4"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) {
        asm volatile(".rept 16\n" I_FMA_VVV(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) ".endr\n" BODY_INPUTS);
    }
};

// transcendental : FMA mixes -- does the transcendental unit overlap with plain VALU issue?
#define MIX(NAME, LABEL, SEQ, COUNT)                                                                          \
    struct NAME {                                                                                             \
        static constexpr int N = COUNT;                                                                       \
        static const char *name() { return LABEL; }                                                           \
        __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept 8\n" SEQ ".endr\n" BODY_INPUTS); } \
    };
MIX(T_mix_1_1, "mix exp:fma 1:1", I_EXP(A0) I_FMA_VVV(A1) I_EXP(A2) I_FMA_VVV(A3) I_EXP(A4) I_FMA_VVV(A5) I_EXP(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_1_3, "mix exp:fma 1:3", I_EXP(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_EXP(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_1_7, "mix exp:fma 1:7", I_EXP(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_sqrt_1_3, "mix sqrt:fma 1:3", I_SQRT(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_SQRT(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_max_fma, "mix max:fma 1:1", I_MAX(A0) I_FMA_VVV(A1) I_MAX(A2) I_FMA_VVV(A3) I_MAX(A4) I_FMA_VVV(A5) I_MAX(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_cvt_fma, "mix cvt_ubyte0:fma 1:1", I_CVT_UB0(A0) I_FMA_VVV(A1) I_CVT_UB0(A2) I_FMA_VVV(A3) I_CVT_UB0(A4) I_FMA_VVV(A5) I_CVT_UB0(A6) I_FMA_VVV(A7), 64)
MIX(T_mix_exp_max, "mix exp:max 1:3", I_EXP(A0) I_MAX(A1) I_MAX(A2) I_MAX(A3) I_EXP(A4) I_MAX(A5) I_MAX(A6) I_MAX(A7), 64)


// generic pattern test: SEQ issued REPT times per body, COUNT VALU in total
#define PAT(NAME, LABEL, REPT, SEQ, COUNT)                                                                    \
    struct NAME {                                                                                             \
        static constexpr int N = COUNT;                                                                       \
        static const char *name() { return LABEL; }                                                           \
        __device__ static __forceinline__ void body(float *x, const Args &g) { asm volatile(".rept " #REPT "\n" SEQ ".endr\n" BODY_INPUTS); } \
    };
PAT(T_e1_f15, "pattern exp x1, fma x15", 4, I_EXP(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7) ALL8(I_FMA_VVV), 64)
PAT(T_e1_f31, "pattern exp x1, fma x31", 2, I_EXP(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV), 64)
PAT(T_e1_f63, "pattern exp x1, fma x63", 1, I_EXP(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV), 64)
PAT(T_e4_f28, "pattern exp x4 clustered, fma x28", 2, I_EXP(A0) I_EXP(A1) I_EXP(A2) I_EXP(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV), 64)
PAT(T_e8_f56, "pattern exp x8 clustered, fma x56", 1, ALL8(I_EXP) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV) ALL8(I_FMA_VVV), 64)
PAT(T_e2_f14, "pattern exp x2 clustered, fma x14", 4, I_EXP(A0) I_EXP(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7) ALL8(I_FMA_VVV), 64)
PAT(T_m1_f3, "pattern max x1, fma x3", 8, I_MAX(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_MAX(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
PAT(T_m1_f7, "pattern max x1, fma x7", 8, I_MAX(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_FMA_VVV(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
PAT(T_m3_f1, "pattern max x3, fma x1", 8, I_MAX(A0) I_MAX(A1) I_MAX(A2) I_FMA_VVV(A3) I_MAX(A4) I_MAX(A5) I_MAX(A6) I_FMA_VVV(A7), 64)
PAT(T_m4_f4, "pattern max x4 then fma x4", 8, I_MAX(A0) I_MAX(A1) I_MAX(A2) I_MAX(A3) I_FMA_VVV(A4) I_FMA_VVV(A5) I_FMA_VVV(A6) I_FMA_VVV(A7), 64)
PAT(T_sg_f1, "pattern fma_sgpr x1, fma x1", 8, I_FMA_SGPR(A0) I_FMA_VVV(A1) I_FMA_SGPR(A2) I_FMA_VVV(A3) I_FMA_SGPR(A4) I_FMA_VVV(A5) I_FMA_SGPR(A6) I_FMA_VVV(A7), 64)
PAT(T_mulsg_mul, "pattern mul_sgpr x1, mul x1", 8, I_MUL_SGPR(A0) I_MUL(A1) I_MUL_SGPR(A2) I_MUL(A3) I_MUL_SGPR(A4) I_MUL(A5) I_MUL_SGPR(A6) I_MUL(A7), 64)
PAT(T_mul_add, "pattern mul x1, add x1 (both VOP2 fast)", 8, I_MUL(A0) I_ADD(A1) I_MUL(A2) I_ADD(A3) I_MUL(A4) I_ADD(A5) I_MUL(A6) I_ADD(A7), 64)
PAT(T_and_fma, "pattern and x1, fma x1", 8, I_AND(A0) I_FMA_VVV(A1) I_AND(A2) I_FMA_VVV(A3) I_AND(A4) I_FMA_VVV(A5) I_AND(A6) I_FMA_VVV(A7), 64)
PAT(T_cmpcnd_fma, "pattern cmp+cndmask+fma x2", 8, I_CMP_CND(A0) I_FMA_VVV(A1) I_FMA_VVV(A2) I_CMP_CND(A4) I_FMA_VVV(A5) I_FMA_VVV(A6), 64)

// packed f32: 4 pair-accumulators
struct T_pk_fma {
    static constexpr int N = 32;
    static const char *name() { return "v_pk_fma_f32 (2 FMAs per lane)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) {
        f2 p0 = {x[0], x[1]}, p1 = {x[2], x[3]}, p2 = {x[4], x[5]}, p3 = {x[6], x[7]}, m = {g.a, g.a}, c = {g.b, g.b};
        asm volatile(".rept 8\nv_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n.endr\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m), "v"(c));
        x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y; x[6] = p3.x; x[7] = p3.y;
    }
};
struct T_pk_mul {
    static constexpr int N = 32;
    static const char *name() { return "v_pk_mul_f32 (2 MULs per lane)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) {
        f2 p0 = {x[0], x[1]}, p1 = {x[2], x[3]}, p2 = {x[4], x[5]}, p3 = {x[6], x[7]}, m = {g.a, g.a};
        asm volatile(".rept 8\nv_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n.endr\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));
        x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y; x[6] = p3.x; x[7] = p3.y;
    }
};
struct T_pk_add {
    static constexpr int N = 32;
    static const char *name() { return "v_pk_add_f32 (2 ADDs per lane)"; }
    __device__ static __forceinline__ void body(float *x, const Args &g) {
        f2 p0 = {x[0], x[1]}, p1 = {x[2], x[3]}, p2 = {x[4], x[5]}, p3 = {x[6], x[7]}, m = {g.b, g.b};
        asm volatile(".rept 8\nv_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n.endr\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m));
        x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y; x[6] = p3.x; x[7] = p3.y;
    }
};

template <class T>
__global__ __launch_bounds__(256) void kern(WaveRec *rec, float *sink, int iters, Args g) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + (float)((threadIdx.x * 8 + i) & 1023) * (1.0f / 4096.0f);
    __syncthreads();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) T::body(x, g);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 123.456f) sink[threadIdx.x] = s;  // keep the chain alive
    if ((threadIdx.x & 63) == 0) {
        WaveRec w;
        w.t0 = t0; w.t1 = t1; w.r0 = r0; w.r1 = r1;
        w.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        w.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
    }
}

static int g_cus = 256;

template <class T>
int run(WaveRec *d_rec, float *d_sink, int waves_per_simd, int iters) {
    const int blocks = g_cus * waves_per_simd;  // 256-thread blocks: 4 waves, one per SIMD
    const int nw = blocks * 4;
    Args g;
    g.a = 1.0000001f; g.b = 1e-7f; g.s = 0.99999f; g.mask = 0x5555555555555555ull;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<T>, dim3(blocks), dim3(256), 0, 0, d_rec, d_sink, iters, g);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<T>, dim3(blocks), dim3(256), 0, 0, d_rec, d_sink, iters, g);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<WaveRec> rec(nw);
    CHECK(hipMemcpy(rec.data(), d_rec, sizeof(WaveRec) * nw, hipMemcpyDeviceToHost));

    // group by the SIMD the wave ran on: xcc[3:0], se_id[15:13], sh_id[12], cu_id[11:8], simd_id[5:4]
    struct Acc { uint64_t t0 = ~0ull, t1 = 0; int waves = 0; };
    std::map<uint32_t, Acc> simd;
    std::vector<double> clocks, wave_cpi;
    for (const WaveRec &w : rec) {
        const uint32_t key = ((w.xcc & 0xf) << 16) | (w.hwid & 0xff30u);
        Acc &a = simd[key];
        a.t0 = std::min(a.t0, w.t0); a.t1 = std::max(a.t1, w.t1); a.waves += 1;
        if (w.r1 > w.r0) clocks.push_back((double)(w.t1 - w.t0) / (double)(w.r1 - w.r0) * 100.0);  // MHz
        wave_cpi.push_back((double)(w.t1 - w.t0) / ((double)iters * T::N));
    }
    std::vector<double> cpi;
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : simd) {
        cpi.push_back((double)(kv.second.t1 - kv.second.t0) / ((double)kv.second.waves * iters * T::N));
        wmin = std::min(wmin, kv.second.waves); wmax = std::max(wmax, kv.second.waves);
    }
    std::sort(cpi.begin(), cpi.end());
    std::sort(clocks.begin(), clocks.end());
    std::sort(wave_cpi.begin(), wave_cpi.end());
    const double winst = (double)nw * iters * T::N;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"valu_per_wave\": %d, \"cycles_per_winst_per_simd\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f}, "
           "\"wave_cycles_per_own_winst_median\": %.3f, \"clock_mhz\": {\"median\": %.0f, \"min\": %.0f, \"max\": %.0f}, "
           "\"simds_seen\": %zu, \"waves_on_a_simd\": [%d, %d], \"wall_ms\": %.4f, \"winst_per_s_wall\": %.4e}\n",
           T::name(), waves_per_simd, iters * T::N, cpi[cpi.size() / 2], cpi.front(), cpi.back(), wave_cpi[wave_cpi.size() / 2],
           clocks.empty() ? 0.0 : clocks[clocks.size() / 2], clocks.empty() ? 0.0 : clocks.front(), clocks.empty() ? 0.0 : clocks.back(),
           simd.size(), wmin, wmax, ms, winst / (ms * 1e-3));
    fflush(stdout);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

template <class T>
int sweep(WaveRec *d_rec, float *d_sink, bool full) {
    const int iters = 256;
    if (full) {
        for (int w : {1, 2, 4, 8})
            if (run<T>(d_rec, d_sink, w, iters)) return 1;
    } else {
        if (run<T>(d_rec, d_sink, 8, iters)) return 1;
    }
    return 0;
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"note\": \"cycles = s_memtime ticks (shader clock); "
           "clock = d(s_memtime)/d(s_memrealtime) * 100 MHz\"}\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    WaveRec *d_rec;
    float *d_sink;
    CHECK(hipMalloc(&d_rec, sizeof(WaveRec) * (size_t)g_cus * 8 * 4));
    CHECK(hipMalloc(&d_sink, 4096));
#define S(T, FULL) if (sweep<T>(d_rec, d_sink, FULL)) return 1;
    const bool set2 = argc > 1 && std::strcmp(argv[1], "set2") == 0;
    if (set2) {
        S(T_mul_sgpr, false) S(T_add_sgpr, false) S(T_fmac_sgpr, false) S(T_mul_lit, false) S(T_fma_neg, false) S(T_fma_abs, false) S(T_mul_omod, false)
        S(T_or, false) S(T_xor, false) S(T_lshr, false) S(T_ashr, false) S(T_sub_u32, false) S(T_and_or, false) S(T_or3, false) S(T_perm, false)
        S(T_rndne, false) S(T_trunc, false) S(T_mul_u24, false) S(T_cvt_u32, false) S(T_mul_hi, false) S(T_min3, false)
        S(T_mov_dpp, false) S(T_add_dpp, false) S(T_sub_dpp, false) S(T_sqrt_f16, false) S(T_exp_f16, false)
        S(T_mix_1_7, false) S(T_e1_f15, true) S(T_e1_f31, false) S(T_e1_f63, false) S(T_e2_f14, false) S(T_e4_f28, true) S(T_e8_f56, true)
        S(T_m1_f3, false) S(T_m1_f7, false) S(T_m3_f1, false) S(T_m4_f4, false) S(T_sg_f1, false) S(T_mulsg_mul, false) S(T_mul_add, false)
        S(T_and_fma, false) S(T_cmpcnd_fma, false)
        return 0;
    }
    S(T_fma_vvv, true) S(T_fma_acc, false) S(T_fmac, false) S(T_fma_sgpr, false) S(T_fma_ss, false) S(T_fma_const, false)
    S(T_fma_clamp, false) S(T_fmaak, false) S(T_mul, true) S(T_mul_e64, false) S(T_add, false) S(T_sub, false)
    S(T_max, true) S(T_min, false) S(T_med3, false) S(T_max3, false)
    S(T_cvt_ub0, false) S(T_cvt_ub2, false) S(T_cvt_i32, false) S(T_cvt_f32_i32, false) S(T_cvt_f32_u32, false) S(T_floor, false) S(T_fract, false)
    S(T_cndmask_vcc, false) S(T_cndmask_s, false) S(T_cmp, false) S(T_cmp_cnd, false)
    S(T_mov, false) S(T_and, false) S(T_lshl, false) S(T_add_u32, false) S(T_add3_u32, false) S(T_lshl_add, false) S(T_mad_u24, false)
    S(T_mul_lo, false) S(T_bfe, false) S(T_ldexp, false)
    S(T_exp, true) S(T_log, false) S(T_sqrt, false) S(T_rsq, false) S(T_rcp, false)
    S(T_cubeid, false) S(T_cubesc, false) S(T_cubema, false)
    S(T_pk_fma, true) S(T_pk_mul, false) S(T_pk_add, false)
    S(T_fma_chain, true) S(T_fma_dist2, true) S(T_fma_dist4, true) S(T_exp_chain, true)
    S(T_mix_1_1, true) S(T_mix_1_3, true) S(T_mix_1_7, true) S(T_mix_sqrt_1_3, false) S(T_mix_max_fma, true) S(T_mix_cvt_fma, false) S(T_mix_exp_max, false)
    CHECK(hipFree(d_rec));
    CHECK(hipFree(d_sink));
    return 0;
}
