// device_grid.hpp — index arithmetic and the fp16 interpolation step of the multi-resolution hash grid (tcnn HashGrid semantics), shared by the forward
// kernels (matnet.hip) and the backward's forward recompute / gradient scatter (backward.hip): one definition, so the recompute reproduces the forward's bits.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cmath>
#include "device_math.hpp"

namespace mr {

// level table of MLPTexture3D's encoder (render_helper.py:64-76): 16 levels, base resolution 16, T = 2^19, per-level scale exp(ln(256)/15)
#define MR_LEVELS 16

struct GridLevels { float scale[MR_LEVELS]; uint32_t res[MR_LEVELS]; uint32_t size[MR_LEVELS]; uint32_t offset[MR_LEVELS]; };

static GridLevels host_levels(uint32_t* total) {
    GridLevels L;
    const float per_level_scale = 1.4472692012786865f;  // fp32(exp(log(4096/16)/15)), render_helper.py:64-76
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (int i = 0; i < MR_LEVELS; i++) {
        float scale = exp2f(i * log2_pls) * 16 - 1.0f;
        uint32_t res = (uint32_t)ceilf(scale) + 1;
        uint64_t dense = (uint64_t)res * res * res;
        uint32_t params = dense > 0x7fffffffull ? 0x7fffffffu : (uint32_t)dense;
        params = (params + 7u) / 8u * 8u;
        if (params > (1u << 19)) params = 1u << 19;
        L.scale[i] = scale; L.res[i] = res; L.size[i] = params; L.offset[i] = offset;
        offset += params;
    }
    if (total) *total = offset;
    return L;
}

MR_DEV uint32_t grid_index(uint32_t size, uint32_t res, uint32_t px, uint32_t py, uint32_t pz) {
    uint32_t stride = 1, index = 0;
    if (stride <= size) { index += px * stride; stride *= res; }
    if (stride <= size) { index += py * stride; stride *= res; }
    if (stride <= size) { index += pz * stride; stride *= res; }
    if (size < stride) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
    return index % size;
}

// (T)(weight * value) of tcnn's interpolation (grid.h): the fp32 product is ROUNDED TO fp32 and then to fp16. Left to itself the compiler turns
// fptrunc(fmul) into v_fma_mixlo_f16 — one rounding of the exact product — whenever it does not happen to pack the multiplication with a neighbour
// (round 2: -fno-vectorize changed 1 feature in ~10^4 by one fp16 ulp against tcnn's two-step rounding); the empty asm pins the two-step form in every build.
// Both features of a table entry at once: the two fp16 roundings in one v_cvt_pk_f16_f32 (round to nearest even, like v_cvt_f16_f32) and the two
// fp16 additions in one v_pk_add_f16 — packed fp16, written out so that the instruction count does not depend on what a vectoriser finds.
MR_DEV __half2 weighted_half2(float w, __half2 v) {
    float p0 = w * __low2float(v), p1 = w * __high2float(v);
    asm("" : "+v"(p0), "+v"(p1));
    return __floats2half2_rn(p0, p1);
}
// The eight table entries of a cell, tcnn's grid_index for every corner (dx, dy, dz) at once — same values as grid_index(size, res, px + dx, py + dy,
// pz + dz), without its per-corner multiplications and its u32 modulo (a ~30-instruction sequence, eight times per level: it was most of the
// encoder's instructions). The level kind is uniform, so the branch is scalar:
//   hashed (res^3 > size, size = 2^19): (px + dx) ^ (py + dy) * P1 ^ (pz + dz) * P2 with the products of the +1 corners formed by one addition
//     (u32 arithmetic wraps, (p + 1) * P = p * P + P), and `% size` = `& (size - 1)`;
//   dense: px + py * res + pz * res^2 plus the corner strides; an index can pass `size` only at the far faces (a corner coordinate equal to res),
//     where it stays below 2 * size (res + res^2 + res^3 < 2 res^3), so `% size` is one conditional subtraction — the modulo is kept behind a
//     branch no wave takes for points inside the unit cube.
MR_DEV void corner_indices(uint32_t size, uint32_t res, const uint32_t pg[3], uint32_t idx[8]) {
    const bool hashed = (uint64_t)res * res * res > (uint64_t)size && (size & (size - 1u)) == 0u;
    const bool dense = (uint64_t)res * res * res <= (uint64_t)size;
    if (hashed) {
        const uint32_t m = size - 1u;
        const uint32_t hx[2] = {pg[0], pg[0] + 1u};
        const uint32_t y0 = pg[1] * 2654435761u, z0 = pg[2] * 805459861u;
        const uint32_t hy[2] = {y0, y0 + 2654435761u}, hz[2] = {z0, z0 + 805459861u};
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) idx[c] = (hx[c & 1u] ^ hy[(c >> 1) & 1u] ^ hz[c >> 2]) & m;
    } else if (dense) {
        const uint32_t r2 = res * res, base = pg[0] + pg[1] * res + pg[2] * r2;
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            uint32_t i = base + (c & 1u) + ((c >> 1) & 1u) * res + (c >> 2) * r2;
            i = i >= size ? i - size : i;
            if (i >= size) i %= size;
            idx[c] = i;
        }
    } else {
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) idx[c] = grid_index(size, res, pg[0] + (c & 1u), pg[1] + ((c >> 1) & 1u), pg[2] + (c >> 2));
    }
}
// The eight entries of a cell in four 8-byte gathers where the table allows it: the corners (dx = 0, 1) of one (dy, dz) are neighbours in memory
// on a dense level (indices i, i + 1) and, on a hashed level, whenever px is even ((px + 1) ^ h = (px ^ h) ^ 1: the two entries of one aligned pair).
// Elsewhere — odd px on a hashed level, the wrap at a dense level's far face — the second entry is fetched by its own (lane-masked) gather.
// Same entries, same values: the encoder's time goes into these gathers (one L1 tag look-up per distinct line per instruction), not into arithmetic.
struct __attribute__((packed, aligned(4))) GridPair { uint32_t x, y; };
MR_DEV __half2 as_half2(uint32_t u) { __half2 h; __builtin_memcpy(&h, &u, 4); return h; }
MR_DEV void gather_cell(const __half2* __restrict__ g, const uint32_t ci[8], bool hashed, __half2 v[8]) {
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        const uint32_t i0 = ci[c], i1 = ci[c + 1];
#ifndef MR_PAIR_MODE
#define MR_PAIR_MODE 3
#endif
        if ((hashed && !(MR_PAIR_MODE & 2)) || (!hashed && !(MR_PAIR_MODE & 1))) { v[c] = g[i0]; v[c + 1] = g[i1]; }
        else if (hashed) {
            const uint2 pr = *reinterpret_cast<const uint2*>(g + (i0 & ~1u));
            const bool odd = (i0 & 1u) != 0u;
            v[c] = as_half2(odd ? pr.y : pr.x);
            uint32_t o = odd ? pr.x : pr.y;
            if ((i0 ^ i1) != 1u) o = *reinterpret_cast<const uint32_t*>(g + i1);
            v[c + 1] = as_half2(o);
        } else {
            const GridPair pr = *reinterpret_cast<const GridPair*>(g + i0);
            v[c] = as_half2(pr.x);
            uint32_t o = pr.y;
            if (i1 != i0 + 1u) o = *reinterpret_cast<const uint32_t*>(g + i1);
            v[c + 1] = as_half2(o);
        }
    }
}
}  // namespace mr
