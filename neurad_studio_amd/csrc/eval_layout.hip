// Eval-time re-layout of the coarse levels of a hash grid (round 3).
//
// HashEncoding hashes EVERY level (encodings.py:419-444), also the ones whose lattice has far fewer points than the table
// has entries: level 0 of BASELINE config[1] (res 16) has 17^3 = 4913 lattice points scattered over 2^19 entries, each in
// a cache line of its own.  For inference the table is frozen, so those levels can be read from a SHADOW copy in which the
// lattice point (ix, iy, iz) sits at row  ix | iy << s | iz << 2s  (s = bits of the level's largest coordinate): the 8
// corners of a cell then lie in 4 lines instead of 8, neighbouring samples share lines, and the level's working set is a
// few hundred KB instead of "one line per lattice point".  Because the three bit fields do not overlap, OR == XOR, so the
// shadow index is   (ix * 1) ^ (iy * 2^s) ^ (iz * 2^2s) & (2^3s - 1)   -- the reference's hash formula with other
// multipliers and another mask: the fused kernel runs ONE code path with per-level constants {mulY, mulZ, mask, row0}.
// Levels whose 2^3s would exceed the table size keep the reference hash (mulY / mulZ = the primes, mask = T - 1).
// The eval table = all levels back to back in that layout; values are copies, so outputs are bit-identical.  It is a
// cache owned by the caller (rebuilt when the parameters change); the state_dict never sees it.
#include "common.h"

namespace nrhip {
namespace {

struct LayoutDev {
  uint32_t lay[NRHIP_MAX_LEVELS * 4];  // {mulY, mulZ, mask, row0} per level
  int L, log2T, elem_bytes;
};

__global__ __launch_bounds__(256) void eval_layout_build_kernel(LayoutDev ld, const unsigned char* __restrict__ table,
                                                                unsigned char* __restrict__ out, int64_t rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  int l = 0;
  while (l + 1 < ld.L && (int64_t)ld.lay[4 * (l + 1) + 3] <= r) ++l;
  const uint32_t local = (uint32_t)(r - ld.lay[4 * l + 3]);
  const uint32_t my = ld.lay[4 * l], mask = ld.lay[4 * l + 2];
  uint32_t src = local;
  if (my != kPrimeY) {  // shadow level: row = ix | iy << s | iz << 2s  ->  the entry the reference's hash points at
    const int s = __ffs((int)my) - 1;
    const uint32_t ix = local & (my - 1u), iy = (local >> s) & (my - 1u), iz = local >> (2 * s);
    src = (ix ^ (iy * kPrimeY) ^ (iz * kPrimeZ)) & ((1u << ld.log2T) - 1u);
    (void)mask;
  }
  const unsigned char* sp = table + (((size_t)l << ld.log2T) + src) * ld.elem_bytes;
  unsigned char* dp = out + (size_t)r * ld.elem_bytes;
  if (ld.elem_bytes % 16 == 0) {
    for (int k = 0; k < ld.elem_bytes; k += 16) *reinterpret_cast<uint4*>(dp + k) = *reinterpret_cast<const uint4*>(sp + k);
  } else if (ld.elem_bytes % 8 == 0) {
    for (int k = 0; k < ld.elem_bytes; k += 8) *reinterpret_cast<uint2*>(dp + k) = *reinterpret_cast<const uint2*>(sp + k);
  } else if (ld.elem_bytes % 4 == 0) {
    for (int k = 0; k < ld.elem_bytes; k += 4) *reinterpret_cast<uint32_t*>(dp + k) = *reinterpret_cast<const uint32_t*>(sp + k);
  } else {
    for (int k = 0; k < ld.elem_bytes; k += 2) *reinterpret_cast<uint16_t*>(dp + k) = *reinterpret_cast<const uint16_t*>(sp + k);
  }
}

}  // namespace
}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_eval_layout_plan(const nrhip_grid* g, uint32_t* layout, int64_t* rows) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(layout && rows, NRHIP_ERR_INVALID_ARG, "eval_layout_plan: NULL output");
  const int64_t T = (int64_t)1 << g->log2_table_size;
  int64_t acc = 0;
  for (int l = 0; l < g->num_levels; ++l) {
    // coordinates of a position in [0,1]^3 at this level: floor / ceil of x * scalings[l], i.e. 0 .. ceil(scalings[l])
    const float sc = g->scalings[l];
    NR_REQUIRE(sc >= 1.f && sc < 16777216.f, NRHIP_ERR_INVALID_ARG, "eval_layout_plan: scalings[%d] = %g", l, sc);
    const uint32_t maxc = (uint32_t)ceilf(sc);
    int s = 1;
    while ((1u << s) <= maxc) ++s;
    const bool shadow = 3 * s <= g->log2_table_size && 3 * s <= 30;
    layout[4 * l + 0] = shadow ? (1u << s) : kPrimeY;
    layout[4 * l + 1] = shadow ? (1u << (2 * s)) : kPrimeZ;
    layout[4 * l + 2] = shadow ? ((1u << (3 * s)) - 1u) : (uint32_t)(T - 1);
    NR_REQUIRE(acc < ((int64_t)1 << 32), NRHIP_ERR_UNSUPPORTED, "eval_layout_plan: more than 2^32 rows");
    layout[4 * l + 3] = (uint32_t)acc;
    acc += shadow ? ((int64_t)1 << (3 * s)) : T;
  }
  NR_REQUIRE(acc * g->n_features * (g->param_dtype == 1 ? 2 : 4) <= ((int64_t)1 << 32), NRHIP_ERR_UNSUPPORTED,
             "eval_layout_plan: eval table larger than 4 GiB");
  *rows = acc;
  return NRHIP_OK;
}

extern "C" int nrhip_eval_layout_build(const nrhip_grid* g, const void* table, const uint32_t* layout, void* eval_table,
                                       void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(table && layout && eval_table, NRHIP_ERR_INVALID_ARG, "eval_layout_build: NULL pointer");
  LayoutDev ld;
  int64_t rows = 0;
  for (int l = 0; l < g->num_levels; ++l) {
    for (int k = 0; k < 4; ++k) ld.lay[4 * l + k] = layout[4 * l + k];
    const uint32_t m = layout[4 * l + 2];
    rows = (int64_t)layout[4 * l + 3] + (int64_t)m + 1;
  }
  ld.L = g->num_levels, ld.log2T = g->log2_table_size;
  ld.elem_bytes = g->n_features * (g->param_dtype == 1 ? 2 : 4);
  eval_layout_build_kernel<<<grid_for(rows, 256), 256, 0, (hipStream_t)stream>>>(
      ld, static_cast<const unsigned char*>(table), static_cast<unsigned char*>(eval_table), rows);
  return check_launch("eval_layout_build");
}
