// Dev micro-benchmark (GPU box): issue cost of single VALU instructions on gfx950 — cycles per wave64 instruction per SIMD at full occupancy,
// independent chains (throughput, not latency).  hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITER 512
#define STR2(x) #x
#define STR(x) STR2(x)
// 8 independent destination registers per body so that dependent-issue latency does not limit the rate
#define BODY8(INS) \
    INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)
#define KERNEL(NAME, INS, ...)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float* out, float a, float b) {                                  \
        float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;         \
        float s = b, u = b * 0.5f; asm volatile("s_mov_b64 s[24:25], -1\n s_mov_b32 s26, 0x3f800000\n s_mov_b64 vcc, -1" ::: "s24", "s25", "s26", "vcc");                                                                            \
        for (int it = 0; it < ITER; it++) {                                                                      \
            asm volatile(".rept " STR(REP) "\n" BODY8(INS) ".endr\n"                                             \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)       \
                         : "v"(s), "v"(u) : __VA_ARGS__);                                                       \
        }                                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                     \
    }
#define I_FMA(d) "v_fma_f32 " #d ", " #d ", %8, %9\n"
#define I_MUL(d) "v_mul_f32 " #d ", " #d ", %8\n"
#define I_ADD(d) "v_add_f32 " #d ", " #d ", %8\n"
#define I_MAX(d) "v_max_f32 " #d ", " #d ", %8\n"
#define I_MAX3(d) "v_max3_f32 " #d ", " #d ", %8, %9\n"
#define I_MIN3(d) "v_min3_f32 " #d ", " #d ", %8, %9\n"
#define I_CVTUB(d) "v_cvt_f32_ubyte1 " #d ", " #d "\n"
#define I_CVTI(d) "v_cvt_f32_i32 " #d ", " #d "\n"
#define I_CND(d) "v_cndmask_b32 " #d ", " #d ", %8, vcc\n"
#define I_CMP(d) "v_cmp_gt_f32 vcc, " #d ", %8\n"
#define I_CMPE64(d) "v_cmp_gt_f32 s[20:21], " #d ", %8\n"
#define I_AND(d) "v_and_b32 " #d ", " #d ", %8\n"
#define I_LSHL(d) "v_lshlrev_b32 " #d ", 3, " #d "\n"
#define I_ADDU(d) "v_add_u32 " #d ", " #d ", %8\n"
#define I_BFE(d) "v_bfe_u32 " #d ", " #d ", 8, 8\n"
#define I_PERM(d) "v_perm_b32 " #d ", " #d ", %8, %9\n"
#define I_RCP(d) "v_rcp_f32 " #d ", " #d "\n"
#define I_SQRT(d) "v_sqrt_f32 " #d ", " #d "\n"
#define I_FMAMIX(d) "v_fma_mix_f32 " #d ", " #d ", %8, %9 op_sel_hi:[1,0,0]\n"
#define I_PKFMA(d) "v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[44:45]\n"
#define I_MOV(d) "v_mov_b32 " #d ", %8\n"
#define I_MAD_U32(d) "v_mad_u32_u24 " #d ", " #d ", %8, %9\n"
#define I_MULLO(d) "v_mul_lo_u32 " #d ", " #d ", %8\n"
#define I_XOR(d) "v_xor_b32 " #d ", " #d ", %8\n"
#define I_ANDOR(d) "v_and_or_b32 " #d ", " #d ", %8, %9\n"
#define I_DPP(d) "v_mov_b32_dpp " #d ", " #d " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_FMAC_DPP(d) "v_fmac_f32_dpp " #d ", %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_LDEXP(d) "v_ldexp_f32 " #d ", " #d ", %8\n"
#define I_CND64(d) "v_cndmask_b32_e64 " #d ", " #d ", %8, s[24:25]\n"
#define I_CNDVCC0(d) "v_cndmask_b32 " #d ", " #d ", %8, vcc\n"
#define I_FMAC(d) "v_fmac_f32 " #d ", %8, %9\n"
#define I_FMA_K(d) "v_fma_f32 " #d ", " #d ", 2.0, %8\n"
#define I_FMA_S(d) "v_fma_f32 " #d ", " #d ", s26, %8\n"
#define I_MIN(d) "v_min_f32 " #d ", " #d ", %8\n"
#define I_SUB(d) "v_sub_f32 " #d ", " #d ", %8\n"
#define I_MADF(d) "v_mad_f32 " #d ", " #d ", %8, %9\n"
#define I_LSHLADD(d) "v_lshl_add_u32 " #d ", " #d ", 2, %8\n"
#define I_OR(d) "v_or_b32 " #d ", " #d ", %8\n"
#define I_CMPLT_I(d) "v_cmp_lt_i32 vcc, " #d ", %8\n"
#define I_CMP_CND(d) "v_cmp_gt_f32 vcc, " #d ", %8\n v_cndmask_b32 " #d ", " #d ", %9, vcc\n"
#define I_MULF_E64(d) "v_mul_f32_e64 " #d ", " #d ", %8\n"
#define I_SALU(d) "s_add_u32 s20, s20, 1\n"
KERNEL(k_fma, I_FMA, "memory") KERNEL(k_mul, I_MUL, "memory") KERNEL(k_add, I_ADD, "memory") KERNEL(k_max, I_MAX, "memory")
KERNEL(k_max3, I_MAX3, "memory") KERNEL(k_min3, I_MIN3, "memory") KERNEL(k_cvtub, I_CVTUB, "memory") KERNEL(k_cvti, I_CVTI, "memory")
KERNEL(k_cnd, I_CND, "vcc") KERNEL(k_cmp, I_CMP, "vcc") KERNEL(k_and, I_AND, "memory") KERNEL(k_lshl, I_LSHL, "memory")
KERNEL(k_addu, I_ADDU, "memory") KERNEL(k_bfe, I_BFE, "memory") KERNEL(k_perm, I_PERM, "memory") KERNEL(k_rcp, I_RCP, "memory") KERNEL(k_sqrt, I_SQRT, "memory")
KERNEL(k_fmamix, I_FMAMIX, "memory") KERNEL(k_mov, I_MOV, "memory") KERNEL(k_madu, I_MAD_U32, "memory") KERNEL(k_mullo, I_MULLO, "memory")
KERNEL(k_cnd64, I_CND64, "memory") KERNEL(k_fmac, I_FMAC, "memory") KERNEL(k_fmak, I_FMA_K, "memory") KERNEL(k_fmas, I_FMA_S, "memory")
KERNEL(k_min, I_MIN, "memory") KERNEL(k_sub, I_SUB, "memory") KERNEL(k_lshladd, I_LSHLADD, "memory") KERNEL(k_or, I_OR, "memory")
KERNEL(k_cmplti, I_CMPLT_I, "vcc") KERNEL(k_cmpcnd, I_CMP_CND, "vcc") KERNEL(k_mule64, I_MULF_E64, "memory")
KERNEL(k_xor, I_XOR, "memory") KERNEL(k_andor, I_ANDOR, "memory") KERNEL(k_dpp, I_DPP, "memory") KERNEL(k_ldexp, I_LDEXP, "memory")

typedef void (*kern_t)(float*, float, float);
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; hipMalloc(&out, 4 * 256 * 256 * 8 * 4);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3}, {"v_min3_f32", k_min3},
        {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_f32_i32", k_cvti}, {"v_cndmask_b32", k_cnd}, {"v_cmp_gt_f32 vcc", k_cmp},
        {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_add_u32", k_addu}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_rcp_f32", k_rcp},
        {"v_sqrt_f32", k_sqrt}, {"v_fma_mix_f32", k_fmamix}, {"v_mov_b32", k_mov}, {"v_mad_u32_u24", k_madu}, {"v_mul_lo_u32", k_mullo}, {"v_xor_b32", k_xor}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"v_fmac_f32", k_fmac}, {"v_fma_f32 d,d,2.0,v", k_fmak}, {"v_fma_f32 d,d,s26,v", k_fmas},
        {"v_min_f32", k_min}, {"v_sub_f32", k_sub}, {"v_lshl_add_u32", k_lshladd}, {"v_or_b32", k_or}, {"v_cmp_lt_i32 vcc", k_cmplti},
        {"v_cmp+v_cndmask (pair)", k_cmpcnd}, {"v_mul_f32_e64", k_mule64},
        {"v_and_or_b32", k_andor}, {"v_mov_b32_dpp", k_dpp}, {"v_ldexp_f32", k_ldexp}};
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    for (int wps : {2, 6}) {   // waves per SIMD
        printf("---- %d wave(s) per SIMD\n", wps);
        for (auto& e : ks) {
            const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            e.k<<<blocks, 256>>>(out, 1.0f, 1.0001f); hipDeviceSynchronize();
            hipEventRecord(a); e.k<<<blocks, 256>>>(out, 1.0f, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr_per_simd = (double)wps * ITER * REP * 8;
            const double cyc = ms * 1e-3 * (double)p.clockRate * 1e3 / instr_per_simd;
            printf("  %-24s %7.3f ms  %5.2f cycles per wave-instruction per SIMD (at %d kHz nominal)\n", e.name, ms, cyc, p.clockRate);
        }
    }
    return 0;
}
