"""Writes the source variants of tools/probes/spill_kit/make.sh next to T/v_fold.hip (the archived, bit-stable form of the
folded merge: every list, the merging slice's own included, is read back from scratch).  usage: variants.py TREE_DIR

  haz          the form that failed in round 5: the merging slice keeps ITS list in registers (2 spilled VGPRs)
  haz_nospill  the same source at 5 instead of 7 workgroups per CU: 0 spilled VGPRs            -> fails the same way
  haz_top/bot  the merging slice is always the highest / lowest one (bounded spin on the arrival word) -> fail
  haz_top_ns / haz_bot_ns   the same without the spilled registers: what repro_lists.py takes apart
  p1_wait0     s_waitcnt vmcnt(0) between the loads of a list and its insertions               -> fails
  p2_nop       s_nop 0 in the same place (control for p1)                                      -> fails
  p3_plain     plain loads instead of agent-scope atomic loads of the other slices' lists      -> fails
  p4_ownfirst  v_fold with the own list inserted FIRST (all lists from scratch)                -> correct
  haz_dbg      haz_nospill + read-only order checks of the lists before / after the merge      -> correct
  haz_post     haz_nospill + a second all-from-scratch merge AFTER the merge, compared         -> correct, 0 differences
"""
import sys

T = sys.argv[1].rstrip("/") + "/"
fold = open(T + "v_fold.hip").read()


def sub(s, a, b, count=1):
    assert s.count(a) >= 1, a[:60]
    return s.replace(a, b, count)


own_from_scratch = """    best.init();
    for (int sI = 0; sI < nslices; ++sI) {
      const float* so = scratch + (int64_t)(b0 + sI) * 3 * KMAX * 256;"""
haz = sub(fold, own_from_scratch, """    for (int sI = 0; sI < nslices; ++sI) {
      if (sI == slice) continue;                         // own list: kept in registers (the hazardous form)
      const float* so = scratch + (int64_t)(b0 + sI) * 3 * KMAX * 256;""")
open(T + "v_haz.hip", "w").write(haz)
nospill = sub(haz, "__launch_bounds__(256, (CP && KMAX <= 8) ? 7 : 1) void k_raster(",
              "__launch_bounds__(256, (CP && KMAX <= 8) ? 5 : 1) void k_raster(")
open(T + "v_haz_nospill.hip", "w").write(nospill)

arrive = """    if (threadIdx.x == 0)
      s_last = __hip_atomic_fetch_add(&arrive[b0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nslices - 1;"""
fixed = """    if (threadIdx.x == 0) {
      if (slice != MERGER_SLICE) { __hip_atomic_fetch_add(&arrive[b0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_last = 0; }
      else {
        int spins = 0;
        while (__hip_atomic_load(&arrive[b0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nslices - 1 && ++spins < 4000000) __builtin_amdgcn_s_sleep(8);
        s_last = 1;
      }
    }"""
open(T + "v_haz_top.hip", "w").write(sub(haz, arrive, fixed.replace("MERGER_SLICE", "(nslices - 1)")))
open(T + "v_haz_bot.hip", "w").write(sub(haz, arrive, fixed.replace("MERGER_SLICE", "0")))
# ... and the same two without the spilled registers (repro_lists.py reads the slices' lists back from the workspace)
open(T + "v_haz_top_ns.hip", "w").write(sub(nospill, arrive, fixed.replace("MERGER_SLICE", "(nslices - 1)")))
open(T + "v_haz_bot_ns.hip", "w").write(sub(nospill, arrive, fixed.replace("MERGER_SLICE", "0")))

pushes = """#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j0 + j < K && zz[j] < FLT_MAX) best.push(zz[j], ii[j], qq[j], K);"""
assert nospill.count(pushes) == 1
open(T + "v_p1_wait0.hip", "w").write(nospill.replace(pushes, '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n' + pushes))
open(T + "v_p2_nop.hip", "w").write(nospill.replace(pushes, '        asm volatile("s_nop 0" ::: "memory");\n' + pushes))
p3 = nospill
for a, b in (("zz[j] = __hip_atomic_load(so + (j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);", "zz[j] = so[(j0 + j) * 256 + threadIdx.x];"),
             ("qq[j] = __hip_atomic_load(so + (KMAX + j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);", "qq[j] = so[(KMAX + j0 + j) * 256 + threadIdx.x];"),
             ("ii[j] = __float_as_int(__hip_atomic_load(so + (2 * KMAX + j0 + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));", "ii[j] = __float_as_int(so[(2 * KMAX + j0 + j) * 256 + threadIdx.x]);")):
    p3 = sub(p3, a, b)
open(T + "v_p3_plain.hip", "w").write(p3)
open(T + "v_p4_ownfirst.hip", "w").write(sub(fold, own_from_scratch, """    best.init();
    for (int sJ = 0; sJ < nslices; ++sJ) {
      const int sI = sJ == 0 ? slice : (sJ <= slice ? sJ - 1 : sJ);
      const float* so = scratch + (int64_t)(b0 + sI) * 3 * KMAX * 256;"""))

counters = """__device__ unsigned g_dbg[16];
__device__ int g_dump[64 * 64];
extern "C" int iso_dbg_counts(unsigned* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dbg), 64) != hipSuccess) return -1;
  unsigned z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, 64) == hipSuccess ? 0 : -1;
}
extern "C" int iso_dbg_dump(int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dump), sizeof(int) * 64 * 64) == hipSuccess ? 0 : -1; }
struct TileItemsJob {"""
merge_loop_head = """    for (int sI = 0; sI < nslices; ++sI) {
      if (sI == slice) continue;                         // own list: kept in registers (the hazardous form)"""
dbg = sub(nospill, "struct TileItemsJob {", counters)
dbg = sub(dbg, merge_loop_head, """    if (threadIdx.x == 0) atomicAdd(&g_dbg[5], 1u);
#pragma unroll
    for (int j = 0; j + 1 < KMAX; ++j) {                 // the own list as it stands in registers
      if (best.z[j + 1] < FLT_MAX) {
        if (best.z[j] > best.z[j + 1]) atomicAdd(&g_dbg[0], 1u);
        if (best.z[j] == best.z[j + 1] && best.id[j] > best.id[j + 1]) atomicAdd(&g_dbg[1], 1u);
        if (best.id[j] == best.id[j + 1]) atomicAdd(&g_dbg[2], 1u);
      }
    }
""" + merge_loop_head)
open(T + "v_haz_dbg.hip", "w").write(dbg)

out_mark = """  if (!inside) return;
  // output pixel is flipped in both axes (+X left, +Y up; rasterize_points.cu:577-580)"""
post = sub(nospill, "struct TileItemsJob {", counters)
post = sub(post, out_mark, """  if (nslices > 1) {
    // AFTER the merge: the same merge again with every list (the own one included) read back from scratch into a fresh
    // list; a pixel whose two results differ dumps both
    const int b0 = slot - slice;
    PixK<KMAX> chk;
    chk.init();
    for (int sI = 0; sI < nslices; ++sI) {
      const float* so = scratch + (int64_t)(b0 + sI) * 3 * KMAX * 256;
      for (int j = 0; j < KMAX; ++j) {
        const float zz = __hip_atomic_load(so + j * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float qq = __hip_atomic_load(so + (KMAX + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ii = __float_as_int(__hip_atomic_load(so + (2 * KMAX + j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (j < K && zz < FLT_MAX) chk.push(zz, ii, qq, K);
      }
    }
    bool bad = false;
    for (int j = 0; j < KMAX; ++j) bad = bad || chk.id[j] != best.id[j];
    if (bad) {
      const unsigned at = atomicAdd(&g_dbg[0], 1u);
      if (at < 64) {
        int* o = g_dump + at * 64;
        o[0] = slice; o[1] = nslices; o[2] = threadIdx.x; o[3] = tile;
        for (int j = 0; j < KMAX; ++j) { o[8 + j] = best.id[j]; o[16 + j] = chk.id[j]; o[24 + j] = __float_as_int(best.z[j]); }
        const float* so = scratch + (int64_t)slot * 3 * KMAX * 256;
        for (int j = 0; j < KMAX; ++j) { o[32 + j] = __float_as_int(so[(2 * KMAX + j) * 256 + threadIdx.x]); o[40 + j] = __float_as_int(so[j * 256 + threadIdx.x]); }
      }
    }
  }
""" + out_mark)
open(T + "v_haz_post.hip", "w").write(post)
print("variants written to", T)
