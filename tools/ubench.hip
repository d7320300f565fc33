// ubench.hip - VALU issue-rate probe for gfx950: how many lane-ops/s the chip sustains for the
// instructions the tracing kernels are made of.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));
#define ITER 4096
#define CHAINS 8

static __device__ inline float tof(float v) { return v; }
static __device__ inline float tof(unsigned v) { return (float)v; }
static __device__ inline float tof(float2_t v) { return v.x + v.y; }
static __device__ inline float2_t mk2(float a, float b) { float2_t v; v.x = a; v.y = b; return v; }
#define KERNEL(NAME, DECL, BODY)                                                              \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {               \
        DECL;                                                                                 \
        for (int i = 0; i < ITER; ++i) {                                                      \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { BODY; }                      \
        }                                                                                     \
        float s = 0;                                                                          \
        for (int c = 0; c < CHAINS; ++c) s += tof(x[c]);                                    \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
    }

KERNEL(k_fma, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
KERNEL(k_mul, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_cndmask, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a)))
KERNEL(k_cmp, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : "+v"(x[c]) : "v"(a) : "vcc"))
KERNEL(k_sqrt, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c + 1, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c])))
KERNEL(k_rcp, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c + 1, asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c])))
KERNEL(k_sin, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = 0.001f * (threadIdx.x + c), asm volatile("v_sin_f32 %0, %0" : "+v"(x[c])))
KERNEL(k_mullo, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_mul24, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_mad24, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_xor, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_lshr, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x[c])))
KERNEL(k_cvt, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[c])))


KERNEL(k_cnd64, float x[CHAINS]; unsigned long long m = __ballot(threadIdx.x & 1); for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "s"(m)))
KERNEL(k_cmpcnd, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[c]) : "v"(a), "v"(b) : "vcc"))
KERNEL(k_min, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_add, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_fmac, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
KERNEL(k_fmas, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "s"(a), "v"(b)))
KERNEL(k_rsq, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c + 1, asm volatile("v_rsq_f32 %0, %0" : "+v"(x[c])))
KERNEL(k_mov, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_mulhi, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_xad, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_med3, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))

KERNEL(k_add_s, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a)))
KERNEL(k_mul_s, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a)))
KERNEL(k_sub_s, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a)))
KERNEL(k_fmac_s, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[c]) : "s"(a), "v"(b)))
KERNEL(k_fma_k, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[c]) : "v"(a)))
KERNEL(k_max, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_cmp_s, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_lt_f32 vcc, %1, %0" : "+v"(x[c]) : "s"(a) : "vcc"))
KERNEL(k_cmp64, float x[CHAINS]; unsigned long long m; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_lt_f32_e64 %1, %0, %2" : "+v"(x[c]), "=s"(m) : "v"(a)))
KERNEL(k_cmpx, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_class_f32 vcc, %0, %1" : "+v"(x[c]) : "v"(a) : "vcc"))
KERNEL(k_and, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_bfi, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_addu, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_lshladd, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_pkmul, float2_t x[CHAINS]; float2_t av = mk2(a, b); for (int c = 0; c < CHAINS; ++c) x[c] = mk2(threadIdx.x, c), asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(av)))
KERNEL(k_pkadd, float2_t x[CHAINS]; float2_t av = mk2(a, b); for (int c = 0; c < CHAINS; ++c) x[c] = mk2(threadIdx.x, c), asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(av)))

__global__ __launch_bounds__(256) void k_clock(float* out, float a, float b) {
    float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER * 4; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) printf("in-kernel clock under v_fma load: %.1f MHz (cycles %llu, realtime ticks %llu)\n", 100.0 * (double)(t1 - t0) / (double)(r1 - r0), t1 - t0, r1 - r0);
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
    float2_t x[CHAINS], av = {a, a}, bv = {b, b};
    for (int c = 0; c < CHAINS; ++c) x[c] = (float2_t){(float)threadIdx.x, (float)c};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(av), "v"(bv));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c].x + x[c].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// overlap probes: a transcendental / compare next to independent cheap VOP2 work (is the cost additive?)
KERNEL(k_sqrt_fmac3, float x[CHAINS]; float y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c + 1; y[c] = c; }, asm volatile("v_sqrt_f32 %0, %0\n\tv_fmac_f32 %1, %2, %3\n\tv_fmac_f32 %1, %3, %2\n\tv_fmac_f32 %1, %2, %2" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b)))
KERNEL(k_cmp_fmac2, float x[CHAINS]; float y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c + 1; y[c] = c; }, asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_fmac_f32 %1, %2, %3\n\tv_fmac_f32 %1, %3, %2" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b) : "vcc"))
KERNEL(k_mul_lit, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(x[c])))
KERNEL(k_cmp_u32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_le_u32 vcc, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a)) : "vcc"))
KERNEL(k_min_u32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_and_or, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_dsread, float x[CHAINS]; __shared__ float4 sm[64]; if (threadIdx.x < 64) sm[threadIdx.x] = make_float4(a, b, a, b); __syncthreads(); unsigned ad = (unsigned)(uintptr_t)sm & 0xffffu; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, { float4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad)); x[c] += v.x; })

KERNEL(k_fmaak, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd" : "+v"(x[c]) : "v"(a)))
KERNEL(k_fmamk, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_min3, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
KERNEL(k_sub_u32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_sub_lit, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_sub_f32 %0, 0x40a00000, %0" : "+v"(x[c])))
KERNEL(k_cnd_vcc2, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[c]) : "v"(a), "v"(b)))
KERNEL(k_cmp_cnd_cnd, float x[CHAINS]; float y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c; y[c] = c; }, asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b) : "vcc"))
KERNEL(k_min_u32_andor, unsigned x[CHAINS]; unsigned y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c; y[c] = c; }, asm volatile("v_and_or_b32 %1, %1, %2, %3\n\tv_min_u32 %0, %0, %1" : "+v"(x[c]), "+v"(y[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_mov_dpp, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[c])))
KERNEL(k_xor_sdwa, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x[c])))
KERNEL(k_add_salu, float x[CHAINS]; int sacc = 0; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1" : "+v"(x[c]), "+s"(sacc) : "v"(a)))
KERNEL(k_add_2salu, float x[CHAINS]; int sacc = 0; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1" : "+v"(x[c]), "+s"(sacc) : "v"(a)))
KERNEL(k_fma_neg, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(x[c]) : "v"(a), "v"(b)))
KERNEL(k_mul_e64neg, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_max_abs, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_max_f32_e64 %0, |%0|, %1" : "+v"(x[c]) : "v"(a)))
KERNEL(k_cvt_ubyte, float x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[c])))
KERNEL(k_cmp_lt_u32_s, unsigned x[CHAINS]; unsigned long long m; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_cmp_lt_u32_e64 %1, %0, %2" : "+v"(x[c]), "=s"(m) : "v"(__float_as_uint(a))))
KERNEL(k_bfe, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x[c])))
KERNEL(k_max_u32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_min_i32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_min3_u32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))
KERNEL(k_add3, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b))))

KERNEL(k_sub_co, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a)) : "vcc"))
KERNEL(k_subco_2cnd, unsigned x[CHAINS]; unsigned y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c; y[c] = c; }, { unsigned t_; asm volatile("v_sub_co_u32_e32 %2, vcc, %0, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %3, %0, vcc\n\tv_cndmask_b32_e32 %1, %4, %1, vcc" : "+v"(x[c]), "+v"(y[c]), "=&v"(t_) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b)) : "vcc"); })
KERNEL(k_cmp_nop_2cnd, float x[CHAINS]; float y[CHAINS]; for (int c = 0; c < CHAINS; ++c) { x[c] = threadIdx.x + c; y[c] = c; }, asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[c]), "+v"(y[c]) : "v"(a), "v"(b) : "vcc"))
KERNEL(k_or_b32, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[c]) : "v"(__float_as_uint(a))))
KERNEL(k_addc, unsigned x[CHAINS]; for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x + c, asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(x[c]) : "v"(__float_as_uint(a)) : "vcc"))

template <typename K>
void run(const char* name, K k, float* d, double ops_per_instr) {
    int blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_instr = 5.0 * blocks * 256.0 * ITER * CHAINS;
    double rate = lane_instr / (ms * 1e-3);
    printf("%-12s %8.3f ms  %7.2f Tlane-instr/s  (%.1f %s)  cycles/wave-instr/SIMD @2.4GHz: %.2f\n", name, ms, rate / 1e12,
           rate * ops_per_instr / 1e12, "Tops/s", 256.0 * 4 * 2.4e9 * 64 / rate);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run("v_fma_f32", k_fma, d, 2); run("v_pk_fma_f32", k_pkfma, d, 4); run("v_mul_f32", k_mul, d, 1);
    run("v_cndmask", k_cndmask, d, 1); run("v_cmp", k_cmp, d, 1); run("v_sqrt_f32", k_sqrt, d, 1); run("v_rcp_f32", k_rcp, d, 1);
    run("v_sin_f32", k_sin, d, 1); run("v_mul_lo_u32", k_mullo, d, 1); run("v_mul_u32_u24", k_mul24, d, 1); run("v_mad_u32_u24", k_mad24, d, 1);
    run("v_xor_b32", k_xor, d, 1); run("v_lshrrev", k_lshr, d, 1); run("v_cvt_f32_u32", k_cvt, d, 1);
    run("cndmask_e64(sgpr)", k_cnd64, d, 1); run("cmp+cndmask", k_cmpcnd, d, 2); run("v_min_f32", k_min, d, 1); run("v_add_f32", k_add, d, 1);
    run("v_fmac_f32", k_fmac, d, 2); run("v_fma sgpr", k_fmas, d, 2); run("v_rsq_f32", k_rsq, d, 1); run("v_mov_b32", k_mov, d, 1);
    run("v_mul_hi_u32", k_mulhi, d, 1); run("v_xad_u32", k_xad, d, 1); run("v_med3_f32", k_med3, d, 1);
    run("v_add_f32 sgpr", k_add_s, d, 1); run("v_mul_f32 sgpr", k_mul_s, d, 1); run("v_sub_f32 sgpr", k_sub_s, d, 1); run("v_fmac sgpr", k_fmac_s, d, 2);
    run("v_fma const", k_fma_k, d, 2); run("v_max_f32", k_max, d, 1); run("v_cmp sgpr", k_cmp_s, d, 1); run("v_cmp_e64->s", k_cmp64, d, 1);
    run("v_cmp_class", k_cmpx, d, 1); run("v_and_b32", k_and, d, 1); run("v_bfi_b32", k_bfi, d, 1); run("v_add_u32", k_addu, d, 1); run("v_lshl_add_u32", k_lshladd, d, 1);
    run("v_pk_mul_f32", k_pkmul, d, 2); run("v_pk_add_f32", k_pkadd, d, 2);
    run("sqrt+3fmac", k_sqrt_fmac3, d, 7); run("cmp+2fmac", k_cmp_fmac2, d, 5); run("v_mul literal", k_mul_lit, d, 1); run("v_cmp_le_u32", k_cmp_u32, d, 1);
    run("v_min_u32", k_min_u32, d, 1); run("v_and_or_b32", k_and_or, d, 1); run("ds_read_b128 same addr (+add)", k_dsread, d, 1);
    run("v_fmaak_f32 (lit)", k_fmaak, d, 2); run("v_fmamk_f32 (lit)", k_fmamk, d, 2); run("v_min3_f32", k_min3, d, 1); run("v_sub_u32", k_sub_u32, d, 1);
    run("v_sub_f32 lit", k_sub_lit, d, 1); run("v_cndmask vcc (no dep)", k_cnd_vcc2, d, 1); run("cmp+2cndmask", k_cmp_cnd_cnd, d, 3); run("and_or+min_u32", k_min_u32_andor, d, 2);
    run("s_nop1+v_mov_dpp", k_mov_dpp, d, 1); run("v_xor_sdwa", k_xor_sdwa, d, 1); run("v_add+s_add", k_add_salu, d, 1); run("v_add+2s_add", k_add_2salu, d, 1);
    run("v_fma neg mod", k_fma_neg, d, 2); run("v_mul_e64 neg", k_mul_e64neg, d, 1); run("v_max_e64 abs", k_max_abs, d, 1); run("v_cvt_f32_ubyte0", k_cvt_ubyte, d, 1);
    run("v_cmp_lt_u32_e64->s", k_cmp_lt_u32_s, d, 1); run("v_bfe_u32", k_bfe, d, 1); run("v_max_u32", k_max_u32, d, 1); run("v_min_i32", k_min_i32, d, 1); run("v_min3_u32", k_min3_u32, d, 1);
    run("v_add3_u32", k_add3, d, 1);
    run("v_sub_co_u32", k_sub_co, d, 1); run("sub_co+nop+2cndmask", k_subco_2cnd, d, 3); run("cmp+nop+2cndmask", k_cmp_nop_2cnd, d, 3); run("v_or_b32", k_or_b32, d, 1); run("v_addc_co_u32", k_addc, d, 1);
    hipLaunchKernelGGL(k_clock, dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipFree(d);
    return 0;
}
