// Issue rate of wave64 vector instructions on one SIMD, by instruction and by the number of waves sharing the SIMD.
// Each wave runs ITER iterations of 16 independent instances of one instruction; a workgroup of 64*W*4 threads puts W waves
// on each of the CU's four SIMDs; one workgroup per CU.  Prints SIMD cycles per instruction (s_memtime ticks of the slowest wave / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2000;

#define OPS16(INS) \
    INS(r0) INS(r1) INS(r2) INS(r3) INS(r4) INS(r5) INS(r6) INS(r7) INS(r8) INS(r9) INS(r10) INS(r11) INS(r12) INS(r13) INS(r14) INS(r15)

#define KERNEL(NAME, ASMSTR) \
__global__ void NAME(unsigned* out, long long* cyc, unsigned a, unsigned b) { \
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, \
             r8 = r0 + 8, r9 = r0 + 9, r10 = r0 + 10, r11 = r0 + 11, r12 = r0 + 12, r13 = r0 + 13, r14 = r0 + 14, r15 = r0 + 15; \
    unsigned va = a + threadIdx.x, vb = b ^ threadIdx.x; \
    __syncthreads(); \
    const long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < ITER; ++i) { \
        OPS16(ASMSTR) \
    } \
    const long long t1 = __builtin_amdgcn_s_memtime(); \
    unsigned s = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ r8 ^ r9 ^ r10 ^ r11 ^ r12 ^ r13 ^ r14 ^ r15; \
    if (s == 0x12345678u) out[threadIdx.x] = s; \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
}

#define I_ADD(r)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_AND(r)   asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_ANDOR(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_PERM(r)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_PKMAX(r) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_PKADD(r) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_PKSUBS(r) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(r) : "v"(va));
#define I_MAX3(r)  asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_MIN16(r) asm volatile("v_min_i16 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_SAD(r)   asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_MAD24(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_MULHI24(r) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_MULLO(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_FMA(r)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(va));
#define I_CMPSEL(r) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r) : "v"(va), "v"(vb) : "vcc");
#define I_BFE(r)   asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r));
#define I_SDWA(r)  asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(r) : "v"(va));
#define I_DPP(r)   asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(va));
#define I_BCNT(r)  asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_CVT(r)   asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r));
#define I_XOR3(r)  asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_MOV(r)   asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(va));
#define I_DOT4(r)  asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r) : "v"(va), "v"(vb));

#define I_LSHL(r)  asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r));
#define I_LSHR(r)  asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r));
#define I_ASHR(r)  asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(r));
#define I_SUB(r)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_OR(r)    asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_XOR(r)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_MAXU(r)  asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_MINI(r)  asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_ADD3(r)  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_LSHLADD(r) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(va));
#define I_ALIGNB(r) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r) : "v"(va));
#define I_DOT2(r)  asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r) : "v"(va), "v"(vb));
#define I_MED3(r)  asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_BITOP3(r) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x40" : "+v"(r) : "v"(va), "v"(vb));
#define I_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(va) : "vcc");
#define I_MUL24(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_PKMULF(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_ADDF(r)  asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(va));
#define I_MIN3(r)  asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(va), "v"(vb));
#define I_CVTI(r)  asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r));

KERNEL(k_lshl, I_LSHL) KERNEL(k_lshr, I_LSHR) KERNEL(k_ashr, I_ASHR) KERNEL(k_sub, I_SUB) KERNEL(k_or, I_OR) KERNEL(k_xor, I_XOR)
KERNEL(k_maxu, I_MAXU) KERNEL(k_mini, I_MINI) KERNEL(k_add3, I_ADD3) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_alignb, I_ALIGNB) KERNEL(k_dot2, I_DOT2)
KERNEL(k_med3, I_MED3) KERNEL(k_bitop3, I_BITOP3) KERNEL(k_cndmask, I_CNDMASK) KERNEL(k_mul24, I_MUL24) KERNEL(k_mulf, I_PKMULF) KERNEL(k_addf, I_ADDF)
KERNEL(k_min3, I_MIN3) KERNEL(k_cvti, I_CVTI)
KERNEL(k_add, I_ADD) KERNEL(k_and, I_AND) KERNEL(k_andor, I_ANDOR) KERNEL(k_perm, I_PERM) KERNEL(k_pkmax, I_PKMAX) KERNEL(k_pkadd, I_PKADD)
KERNEL(k_pksubs, I_PKSUBS) KERNEL(k_max3, I_MAX3) KERNEL(k_min16, I_MIN16) KERNEL(k_sad, I_SAD) KERNEL(k_mad24, I_MAD24) KERNEL(k_mulhi24, I_MULHI24)
KERNEL(k_mullo, I_MULLO) KERNEL(k_fma, I_FMA) KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_cmpsel, I_CMPSEL) KERNEL(k_bfe, I_BFE) KERNEL(k_sdwa, I_SDWA)
KERNEL(k_dpp, I_DPP) KERNEL(k_bcnt, I_BCNT) KERNEL(k_cvt, I_CVT) KERNEL(k_xad, I_XOR3) KERNEL(k_mov, I_MOV) KERNEL(k_dot4, I_DOT4)

struct Entry { const char* name; void (*fn)(unsigned*, long long*, unsigned, unsigned); int per; };

int main()
{
    Entry tab[] = { {"v_add_u32", k_add, 1}, {"v_and_b32", k_and, 1}, {"v_and_or_b32", k_andor, 1}, {"v_perm_b32", k_perm, 1}, {"v_pk_max_i16", k_pkmax, 1},
                    {"v_pk_add_u16", k_pkadd, 1}, {"v_pk_sub_i16 clamp", k_pksubs, 1}, {"v_max3_i32", k_max3, 1}, {"v_min_i16", k_min16, 1}, {"v_sad_u8", k_sad, 1},
                    {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_hi_u32_u24", k_mulhi24, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_fma_f32", k_fma, 1},
                    {"v_lshl_or_b32", k_lshlor, 1}, {"v_cmp+v_cndmask", k_cmpsel, 2}, {"v_bfe_u32", k_bfe, 1}, {"v_add_u32_sdwa", k_sdwa, 1},
                    {"v_add_u32_dpp", k_dpp, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_cvt_f32_ubyte0", k_cvt, 1}, {"v_xad_u32", k_xad, 1}, {"v_mov_b32", k_mov, 1},
                    {"v_dot4_u32_u8", k_dot4, 1},
                    {"v_lshlrev_b32", k_lshl, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_sub_u32", k_sub, 1}, {"v_or_b32", k_or, 1},
                    {"v_xor_b32", k_xor, 1}, {"v_max_u32", k_maxu, 1}, {"v_min_i32", k_mini, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshladd, 1},
                    {"v_alignbyte_b32", k_alignb, 1}, {"v_dot2_u32_u16", k_dot2, 1}, {"v_med3_u32", k_med3, 1}, {"v_bitop3_b32", k_bitop3, 1},
                    {"v_cndmask_b32", k_cndmask, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_mul_f32", k_mulf, 1}, {"v_add_f32", k_addf, 1},
                    {"v_min3_i32", k_min3, 1}, {"v_cvt_f32_i32", k_cvti, 1} };
    unsigned* d_out; long long* d_cyc;
    const int blocks = 256;
    CHECK(hipMalloc(&d_out, 4096 * 4));
    CHECK(hipMalloc(&d_cyc, sizeof(long long) * blocks * 16));
    printf("%-22s", "SIMD cycles/instr at");
    for (int W : {1, 2, 4}) printf("  %d wave%s/SIMD", W, W > 1 ? "s" : " ");
    printf("   (s_memtime ticks of 100 MHz * clock ratio: see ratio line)\n");
    for (const Entry& e : tab) {
        printf("%-22s", e.name);
        for (int W : {1, 2, 4}) {
            const int threads = 64 * 4 * W;
            std::vector<long long> h(blocks * 4 * W);
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 3u, 5u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, 3u, 5u);
            CHECK(hipEventRecord(b));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(h.data(), d_cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double ticks = (double)h[h.size() / 2];
            const double instr_per_simd = (double)ITER * 16 * e.per * W;
            // s_memtime counts at a fixed 100 MHz on this part; report nanoseconds per instruction per SIMD from the event time too
            printf("  %6.2f t (%5.2f ns)", ticks / instr_per_simd, ms * 1e6 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
