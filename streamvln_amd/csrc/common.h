// Shared device helpers for the StreamVLN HIP engine (gfx950 / CDNA4 only).
//
// Every kernel is written once, generic over the storage type T in {bf16, float}:
//   * bf16  -- the shipping compute type (bf16 storage, fp32 accumulate, MFMA 32x32x16 bf16)
//   * float -- parity mode (fp32 storage, MFMA 32x32x2 f32 = exact fp32 fma chain), used to
//              meet the north-star tolerance (token ids identical, hidden states <= 1e-3)
//              against the fp32 CPU oracle.
// The common currency is the 16-byte chunk (uint4): 8 bf16 or 4 floats.  All row strides and
// K extents are multiples of one chunk, so staging / LDS layouts are byte-identical for both
// types and only the MFMA issue differs (mma_chunk below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SVLN_DEV __device__ __forceinline__

struct fp8_t { uint8_t v; };        // OCP e4m3 storage (operands of the opt-in fp8 MFMA products; never an output type)
template <typename T> struct Elt;
template <> struct Elt<fp8_t> { static constexpr int PER_CHUNK = 16; static constexpr int BYTES = 1; };
template <> struct Elt<bf16> { static constexpr int PER_CHUNK = 8; static constexpr int BYTES = 2; };
template <> struct Elt<float> { static constexpr int PER_CHUNK = 4; static constexpr int BYTES = 4; };

SVLN_DEV float to_f32(bf16 v) { return (float)v; }
SVLN_DEV float to_f32(float v) { return v; }
template <typename T> SVLN_DEV T from_f32(float v);
template <> SVLN_DEV bf16 from_f32<bf16>(float v) { return (bf16)v; }    // v_cvt_pk_bf16_f32: RNE, NaN-safe
template <> SVLN_DEV float from_f32<float>(float v) { return v; }

SVLN_DEV uint4 zero_chunk() { return make_uint4(0u, 0u, 0u, 0u); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// streaming (read-once) 16-byte load: non-temporal so weight bytes do not displace reusable lines
SVLN_DEV uint4 load_nt(const void* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// unpack one 16-byte chunk to floats (8 for bf16, 4 for float)
template <typename T> SVLN_DEV void chunk_to_f32(const uint4& c, float* out);
template <> SVLN_DEV void chunk_to_f32<bf16>(const uint4& c, float* out) {
    out[0] = __uint_as_float(c.x << 16); out[1] = __uint_as_float(c.x & 0xFFFF0000u);
    out[2] = __uint_as_float(c.y << 16); out[3] = __uint_as_float(c.y & 0xFFFF0000u);
    out[4] = __uint_as_float(c.z << 16); out[5] = __uint_as_float(c.z & 0xFFFF0000u);
    out[6] = __uint_as_float(c.w << 16); out[7] = __uint_as_float(c.w & 0xFFFF0000u);
}
template <> SVLN_DEV void chunk_to_f32<float>(const uint4& c, float* out) {
    out[0] = __uint_as_float(c.x); out[1] = __uint_as_float(c.y);
    out[2] = __uint_as_float(c.z); out[3] = __uint_as_float(c.w);
}

SVLN_DEV uint32_t pack_bf16x2(float lo, float hi) {
    bf16 a = (bf16)lo, b = (bf16)hi;
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

template <typename T> SVLN_DEV uint4 f32_to_chunk(const float* in);
template <> SVLN_DEV uint4 f32_to_chunk<bf16>(const float* in) {
    return make_uint4(pack_bf16x2(in[0], in[1]), pack_bf16x2(in[2], in[3]), pack_bf16x2(in[4], in[5]),
                      pack_bf16x2(in[6], in[7]));
}
template <> SVLN_DEV uint4 f32_to_chunk<float>(const float* in) {
    return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}

// One "macro" matrix step on 16-byte fragments.  Lane l = (r = l & 31, h = l >> 5) holds chunk
// (2s + h) of row r of both operands (K-contiguous rows: an NT product).
//   bf16 : one v_mfma_f32_32x32x16_bf16   (lane half h holds k = 8h + j, j = 0..7)
//   float: four v_mfma_f32_32x32x2_f32    (MFMA j pairs k = j [h=0] with k = 4 + j [h=1])
// D layout (both): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <typename T> SVLN_DEV void mma_chunk(const uint4& a, const uint4& b, f32x16& acc);
template <> SVLN_DEV void mma_chunk<bf16>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> SVLN_DEV void mma_chunk<float>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// e4m3 operands: a 16-byte chunk holds 16 k values = two v_mfma_f32_32x32x16_fp8_fp8 (8 bytes per lane each); both operands use the
// same byte -> k assignment, so the products pair up correctly whatever the order inside the chunk.  Same rate as the bf16 MFMA,
// half the operand bytes per k.
template <> SVLN_DEV void mma_chunk<fp8_t>(const uint4& a, const uint4& b, f32x16& acc) {
    const long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
    const long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc, 0, 0, 0);
}

SVLN_DEV int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

SVLN_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
SVLN_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activation functions (fp32).  sigmoid on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each: a few ulp of fp32 in all, far
// inside the 1e-3 parity bar) instead of libm's expf / tanhf and an IEEE division: an epilogue evaluates 64-128 activations per lane per
// 256 x 256 tile, and with libm (~35-50 instructions each) that was 35 of the 90 us of the nine-frame fc1 product and ~10 % of gate/up at
// T = 1952.  gelu_tanh uses 0.5 (1 + tanh u) = sigmoid(2u).
SVLN_DEV float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
SVLN_DEV float gelu_tanh_f(float x) {      // ACT2FN["gelu_pytorch_tanh"]  (siglip_encoder.py:83)
    const float k2 = 2.0f * 0.7978845608028654f;   // 2 sqrt(2/pi)
    return x * sigmoid_f(k2 * (x + 0.044715f * x * x * x));
}
SVLN_DEV float gelu_erf_f(float x) {       // nn.GELU()  (multimodal_projector/builder.py:45)
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
SVLN_DEV float silu_f(float x) { return x * sigmoid_f(x); }

// ---- synthetic weights: identical arithmetic to streamvln_amd/weights.py ---------------------
SVLN_DEV uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
SVLN_DEV float synth_value(uint64_t seed_t, uint64_t idx, float step, float base) {
    uint64_t z = splitmix64(seed_t + idx * 0x9E3779B97F4A7C15ull);
    float m = (float)(uint32_t)(z >> 40);
    float v = __fmul_rn(__fsub_rn(m, 8388608.0f), step);
    return base != 0.0f ? __fadd_rn(base, v) : v;
}
