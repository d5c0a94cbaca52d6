// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libyat_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#include "../../include/yat_hip.h"      // status codes (YAT_OK, YAT_EINVAL, ...) and the entry-point prototypes
#define YAT_LDS __attribute__((address_space(3)))

// Tuning switches.  The product library reads NO environment variable and keeps no mutable policy state (include/yat_hip.h,
// "ABI rules"): every YAT_TUNE_* site below is the constant default there.  A tuning build (-DYAT_TUNING, made by
// scripts/build_variant.py for same-box A/B runs through YAT_HIP_LIB) reads the named variable once instead.
#include <stdlib.h>
#ifdef YAT_TUNING
#define YAT_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define YAT_TUNE_F64(name, dflt) (getenv(name) ? atof(getenv(name)) : (dflt))
#else
#define YAT_TUNE_INT(name, dflt) (dflt)
#define YAT_TUNE_F64(name, dflt) (dflt)
#endif

#define YAT_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserving (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// round a float through bf16 (mimics an op boundary of the reference's bf16 dtype flow)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ void unpack8(const u32x4& v, float* o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(v[i] << 16);
        o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (uint32_t)f2bf(f[2 * i]) | ((uint32_t)f2bf(f[2 * i + 1]) << 16);
    return v;
}
__device__ __forceinline__ void unpack4(const u32x2& v, float* o) {
    o[0] = __uint_as_float(v[0] << 16);
    o[1] = __uint_as_float(v[0] & 0xffff0000u);
    o[2] = __uint_as_float(v[1] << 16);
    o[3] = __uint_as_float(v[1] & 0xffff0000u);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
// 4 packed bf16 -> two channel pairs
__device__ __forceinline__ void unpack22(const u32x2& v, f32x2* o) {
    o[0] = f32x2{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u)};
    o[1] = f32x2{__uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    u32x2 v;
    v[0] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
    v[1] = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: results are rounded to bf16 anyway
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = s * (1 + x * (1 - s))
__device__ __forceinline__ float dsilu_from_sigmoid(float x, float s) { return s * (1.0f + x * (1.0f - s)); }
__device__ __forceinline__ float dsilu_f(float x) { return dsilu_from_sigmoid(x, sigmoid_f(x)); }
// GELU(approximate='tanh') through the identity 0.5 (1 + tanh u) = sigmoid(2u), u = k0 (x + k1 x^3): one v_exp_f32 and one
// v_rcp_f32 instead of libm's tanhf (~40 instructions with branches -- in the epilogue of the K = 1152 FFN GEMM of PixArt
// that was half of the main loop's time).  Same function, fp32 rounding differences ~1e-7, far below the bf16 output.
__device__ __forceinline__ float gelu_gate_f(float x) {          // sigmoid(2u)
    const float k0x2 = 2.0f * 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = (k0x2 * x) * __builtin_fmaf(k1 * x, x, 1.0f);
    return fast_rcp(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u2));
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return x * gelu_gate_f(x); }
// d/dx = s + x s' with s = sigmoid(2u): s' = 2 s (1 - s) u',  u' = k0 (1 + 3 k1 x^2)
__device__ __forceinline__ float dgelu_tanh_f(float x) {
    const float k0x2 = 2.0f * 0.7978845608028654f, k1x3 = 3.0f * 0.044715f;
    const float s = gelu_gate_f(x);
    return s * __builtin_fmaf(x * (1.0f - s), k0x2 * __builtin_fmaf(k1x3 * x, x, 1.0f), 1.0f);
}

// Workgroups are dealt round-robin to the 8 XCDs (private L2 each) by their linear index (x fastest).  This maps the
// launch's linear index to a work unit so that every XCD owns one contiguous run of units: neighbouring units -- which
// usually touch neighbouring bytes (the other half of a 128-B line, a halo row) -- then share an L2 instead of each
// fetching their own copy from HBM.  Returns the unit's (x, y, z) in the kernel's own grid coordinates.
__device__ __forceinline__ void xcd_contiguous3(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = u % gx;
    by = (u / gx) % gy;
    bz = u / (gx * gy);
}

// The same for a grid (x = blocks of one (y, z) pair, y, z) whose per-pair work depends on z only (attention over ragged key
// counts: x = query blocks, y = head, z = image): units ordered x fastest, then z, then y.  An XCD's contiguous run then
// holds whole pairs (their x blocks read the same K / V: one L2 fetch instead of eight) AND every image about equally often
// (total / 8 units = a few heads x ALL images), so ragged lengths do not pile up on one XCD as they do with z slowest.
__device__ __forceinline__ void xcd_contiguous3_zfast(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = u % gx;
    bz = (u / gx) % gz;
    by = u / (gx * gz);
}

// ---- MFMA 16x16x32 bf16 fragment loaders (wave64) -------------------------------------------
// Operand register layout (both A and B): lane l holds index idx = l & 15 (row of A / col of B)
// and k = 8*(l>>4) + j, j = 0..7.  Result: lane holds D[row = 4*(l>>4) + r][col = l & 15].

// LDS-DMA: one wave-instruction moves 64 x 16 B = 1 KiB; LDS destination = wave-uniform base +
// lane*16 (linear), global source per lane through a buffer descriptor (out-of-range -> zeros).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, YAT_LDS void* lds_wave_base, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_wave_base, 16, voff, 0, 0, 0);
}
// same, with the uniform part of the address in the instruction's scalar offset (NOT covered by the buffer range check)
__device__ __forceinline__ void lds_dma16s(__amdgpu_buffer_rsrc_t rsrc, YAT_LDS void* lds_wave_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_wave_base, 16, voff, soff, 0, 0);
}
#define YAT_OOB 0x80000000u
// cache-policy operand of the raw buffer builtins: bit 1 = nt (non-temporal).  For data that is written once and next read
// milliseconds later (activations kept for the backward, weight gradients, optimizer state): it should not push the
// operands the next kernels re-read out of the 256 MB Infinity Cache.  Measured in the step, each on its own, three rounds on
// one box (bit-identical results): conv output u -0.28 ms, GEMM pre-activation (aux) stores -0.42, weight-gradient stores
// -0.27, clip + AdamW streams -0.33.
#define YAT_AUX_NT 2

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint64_t bytes) {
    uint32_t n = bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n, 0x00020000);
}

// row image with 128-B rows (64 bf16): 16-B chunk c of row r lives at chunk c ^ ((r>>1)&7)
__device__ __forceinline__ uint32_t swz128(uint32_t row, uint32_t chunk) { return chunk ^ ((row >> 1) & 7); }
// row image with 256-B rows (128 bf16), read by ds_read_b128: chunk c -> c ^ (r & 15)
__device__ __forceinline__ uint32_t swz256(uint32_t row, uint32_t chunk) { return chunk ^ (row & 15); }

__device__ __forceinline__ bf16x8 lds_read8(const char* base, uint32_t byte_off) {
    return *reinterpret_cast<const bf16x8*>(base + byte_off);
}
// transposed 4x16 block read (ds_read_b64_tr_b16): lane 4q+p of each 16-lane group supplies the
// address of row q, cols 4p..4p+3; lane i receives column i of the 4 rows.
__device__ __forceinline__ bf16x4 lds_read_tr4(const char* base, uint32_t byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((YAT_LDS bf16x4*)(base + byte_off));
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
