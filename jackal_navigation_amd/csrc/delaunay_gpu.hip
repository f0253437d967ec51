// delaunay_gpu.hip — the hull recursion of the support points' Delaunay triangulation on the GPU (round 5).  Product code.
//
// What it replaces: csrc/delaunay.cpp's conquer() / zip() on the host — the last per-frame serial code of the path, one pool task per frame
// side (0.66 ms of a core each at 720p; ~10 busy cores at 20 k pairs/s, the part of the path that does not shard with the GPUs).  The
// reference is Shewchuk's Triangle (src/elas/triangle.cpp, called by src/elas/elas.cpp:445-505 as triangulate("zQB")): divide and conquer
// with alternating cuts; support points sit on a lattice, so the triangulation is not unique and every tie-break of that recursion
// shows in the output.  delaunay.cpp is an exact integer replay of it (decision sequence: triangle.cpp:5638-5947 base cases and hull
// zipping, :5953-6103 recursion, :6105-6148 / :7832-7843 output order; predicates exact in int64 = the sign of :2706-2745, :3334-3379).
// This file runs the SAME functions — zip() and the base cases below are delaunay.cpp's, statement for statement — one workgroup per
// frame side, with the whole triangle structure in LDS:
//   * k_arrange (kernels.hip) has already left Triangle's alternating-cut arrangement of the vertices; what remains is the recursion
//     conquer(a, n, axis): a binary tree whose node (depth k, index j) owns the range the repeated halving gives it.  The tree is walked
//     LEVEL BY LEVEL from the leaves up, every node of a level on its own thread (they touch disjoint triangles), a barrier between levels.
//   * Triangle NUMBERS are part of the result (creation order decides which triangle owns a doubly covered pixel downstream).  A merge
//     creates exactly two triangles (base and cap), a leaf two (n = 2) or four (n = 3), so the slot a node's triangles get in the sequential
//     recursion is a function of the sizes alone: count(m) = count(m >> 1) + count(m - (m >> 1)) + 2; a node finds its first slot by
//     walking down from the root.  Output = non-ghost triangles in slot order (a block-wide compaction).
//   * 16-bit everything: handles (triangle << 2 | edge, < 4 x 2n), vertices (ghost = -1), coordinates: 42 bytes per vertex, 3 700
//     vertices per side in 152 KB (a 720p frame has 3 100 - 3 400).  Sides that do not fit, or whose vertices coincide (k_arrange hands
//     those back: which duplicate survives depends on Triangle's randomised quicksort), set the frame's need_host flag and the slot's
//     worker sends the batch through the host stage instead (jn_api.cpp).
//   * Round 6 — sides beyond one workgroup's LDS (1920x1080: ~11 200 support points a side): the tree is cut at depth C.
//     k_delaunay_sub runs every subtree below the cut in its own workgroup (the structure of ITS vertices in LDS, 16-bit links relative to
//     the subtree's first slot) and writes its records, with 32-bit links, and its hull handles to a global scratch; k_delaunay_top runs the
//     C levels above the cut — 2^C - 1 merges — on that global structure (one thread a merge, a memory round trip where the LDS form has an
//     LDS one: slow, but there are few of them) and compacts the output.  The same zip() / leaf(), instantiated for the two memories.
//   * Vertices are numbered by their POSITION in k_arrange's arrangement (round 6; they were the list's indices): the recursion only ever
//     compares vertex numbers and looks their coordinates up, a subtree's vertices are then a contiguous range, and the arrangement itself
//     is needed again only when the triangles are written out (corner = arrangement[position]).
// Time: the top merges are one thread each walking a seam of ~sqrt(n) steps through LDS (a lone wave issues one instruction every ~4.5
// cycles whatever its kind: profiles/r06_lone_wave_probe.txt — the walk costs its instruction count); see DESIGN.md section 8 for the
// measured per-level times.  All sides of a batch side by side on 64 CUs, no host round trip, no host cores.
#include "kernels.h"
#include "hooks.h"
#include <algorithm>
#include <mutex>
#include <type_traits>

namespace jnav {

namespace {

#define DEV static __device__ __forceinline__

// LDS pointers carry their address space: a plain (generic) pointer that is also volatile makes every access a FLAT instruction — through the
// vector-memory path, several hundred nanoseconds each, which is what the first version of this kernel spent its 1.5 ms on.
typedef __attribute__((address_space(3))) const int16_t lds_ci16;
#ifdef JN_DT_NO_VOLATILE      // (probe build: scripts/probes/dt_no_volatile.sh)
#define JN_DT_VOL
#else
#define JN_DT_VOL volatile
#endif
// Where the triangle records live.  LDS: 16-bit links, stored RELATIVE to the first record the workgroup holds (`off`: a subtree's handles
// then fit sixteen bits whatever its place in the whole numbering); global: 32-bit links, absolute.
struct LdsMem {
  typedef __attribute__((address_space(3))) JN_DT_VOL uint16_t* LinkP;
  typedef __attribute__((address_space(3))) JN_DT_VOL int16_t* VertP;
  typedef __attribute__((address_space(3))) JN_DT_VOL uint16_t* HullP;
  typedef uint16_t link_t;
  static constexpr bool kGlobal = false;
  // Round 6: the fourth halfword of every [4]-record is free (it exists so that a handle IS its index), two triangles a vertex: eight bytes a
  // vertex, exactly what its coordinates and its node's two hull handles take.  They live THERE — X(v) in LINK[8 v + 3], Y(v) in LINK[8 v + 7],
  // HL / HR of the node that starts at v in VERT[8 v + 3] / [8 v + 7] — so a side costs 32 bytes a vertex instead of 40 and a CU that holds a
  // k_delaunay workgroup of a 720p side (3 400 vertices: 109 KB) still takes one k_dense_row workgroup (33 KB) beside it.
  static constexpr int kAuxShift = 3;                      // index of vertex v's / position p's auxiliary halfword: (v << 3) [+ 4]
};
// (plain pointers in global memory: a thread reads back what it wrote itself — one wave's accesses to an address keep their order — and what
// other threads wrote reaches it across the fence + barrier between tree levels; volatile would make every access a system-scope round trip)
struct GlobalMem {
  typedef __attribute__((address_space(1))) uint32_t* LinkP;
  typedef __attribute__((address_space(1))) int16_t* VertP;
  typedef __attribute__((address_space(1))) uint32_t* HullP;
  typedef uint32_t link_t;
  static constexpr bool kGlobal = true;
  static constexpr int kAuxShift = 0;                      // coordinates in plain LDS arrays, hull handles in plain global ones
};
template <class M, bool WIDE = false>
struct DT {
  lds_ci16* X; lds_ci16* Y;                 // vertex coordinates (always in LDS), indexed by vertex number - voff
  // volatile: every access is the 16-bit LDS operation it says, in program order.  Round 5 added it after an intermediate version with 6-byte
  // records gave wrong two-vertex leaves once the compiler had merged neighbouring halfword stores.  Looked into in round 6:
  //  * NOT the hardware: ds_write_b16 / b32 / b64 / b96 / b128 land correctly at every byte offset 0..15 on gfx950, and the exact pattern
  //    (two fresh() records of three halfwords through a plain pointer, which the compiler turns into ds_write_b16 + ds_write_b96 at 2-byte
  //    alignment) comes out right (scripts/probes/lds_misaligned_store_probe.hip, profiles/r06_lds_misaligned_store_probe.txt);
  //  * not this tree: with the 8-byte records below a build WITHOUT volatile (-DJN_DT_NO_VOLATILE: 97 ds_write_b16 + 26 ds_write_b32 instead
  //    of 142 + 8) passes tests/test_gpu_delaunay.py and runs in the same time (757 against 760 us a batch, profiles/r06_dt_no_volatile.txt).
  // So the fault was in that intermediate version, not in a misaligned-store rule; volatile stays because it costs nothing here and keeps the
  // access widths what the source says (scripts/probes/dt_no_volatile.sh repeats the comparison after a compiler update).
  typename M::LinkP LINK;                   // [4 T] (the fourth of a record unused): handle across that edge
  typename M::VertP VERT;                   // [4 T]: vertex or -1 (ghost corner)
  uint32_t off;                             // handle of the first record held here (4 x its triangle number); 0 for a whole side and in global memory
  int voff;                                 // number of the first vertex whose coordinates are held here
  int budget;                               // loop iterations left before the node gives up (a corrupt structure must not spin for ever on the GPU; the frame then goes to the host)
  typedef uint32_t H;                       // oriented triangle handle: (triangle << 2) | edge, triangle = its number in the whole side's creation order
  typedef typename M::link_t link_t;
  struct Ctx { int next; };

  __device__ __forceinline__ uint32_t at(H h) const { return h - off; }                      // records are [triangle][4]: the handle IS the index — one address instruction per access
  __device__ __forceinline__ static unsigned up(unsigned e) { return e == 2 ? 0u : e + 1u; }
  __device__ __forceinline__ static unsigned down(unsigned e) { return e == 0 ? 2u : e - 1u; }
  __device__ __forceinline__ static H ccw_edge(H h) { return (h & ~3u) | up(h & 3); }
  __device__ __forceinline__ static H cw_edge(H h) { return (h & ~3u) | down(h & 3); }
  __device__ __forceinline__ H across(H h) const { return (H)LINK[at(h)] + off; }
  __device__ __forceinline__ int org(H h) const { return VERT[at(ccw_edge(h))]; }
  __device__ __forceinline__ int dest(H h) const { return VERT[at(cw_edge(h))]; }
  __device__ __forceinline__ int apex(H h) const { return VERT[at(h)]; }
  __device__ __forceinline__ void set_org(H h, int v) { VERT[at(ccw_edge(h))] = (int16_t)v; }
  __device__ __forceinline__ void set_dest(H h, int v) { VERT[at(cw_edge(h))] = (int16_t)v; }
  __device__ __forceinline__ void set_apex(H h, int v) { VERT[at(h)] = (int16_t)v; }
  __device__ __forceinline__ void glue(H a, H b) { LINK[at(a)] = (link_t)(b - off); LINK[at(b)] = (link_t)(a - off); }
  __device__ __forceinline__ H fresh(Ctx& c) {
    const int t = c.next++;
    const uint32_t i = 4u * (uint32_t)t - off;
    LINK[i] = (link_t)~0u; LINK[i + 1] = (link_t)~0u; LINK[i + 2] = (link_t)~0u;
    VERT[i] = -1; VERT[i + 1] = -1; VERT[i + 2] = -1;
    return (H)t << 2;
  }
  // Predicates in FP64 (WIDE = false).  Coordinates lie in (-2048, 2048) (launch_delaunay's caller says so: `wide` otherwise), so differences
  // are below 2^12, the 2x2 determinants and the squared lengths below 2^25, their products below 2^50 and the sum of three below 2^52:
  // every intermediate is an integer a double holds exactly, i.e. the sign is delaunay.cpp's int64 sign — at a tenth of the instructions
  // (a 64-bit integer multiply is a dozen 32-bit operations here; the top merges are ONE thread, which issues an instruction every ~8 cycles).
  // WIDE (images of 2048 columns or rows and more, coordinates below 8192 — round 6, VERDICT r05 #7 "do not assume it"): differences reach
  // 2^14, the determinants and squared lengths 2^29, their products 2^58 — past what a double holds exactly, so a rounded term could turn
  // the sign of a near-zero sum.  Those handles take integer predicates: 32-bit for `orient` and for in_circle's six small terms, 64-bit
  // products and sum (below 2^60) — delaunay.cpp's own arithmetic, at ~15 more instructions per in_circle than the FP64 form.
  typedef typename std::conditional<WIDE, int32_t, double>::type coord_t;
  struct P { int v; coord_t x, y; };
  __device__ __forceinline__ int xx(int v) const { return X[(v - voff) << M::kAuxShift]; }
  __device__ __forceinline__ int yy(int v) const { return Y[(v - voff) << M::kAuxShift]; }
  __device__ __forceinline__ P pt(int v) const { return P{v, (coord_t)xx(v), (coord_t)yy(v)}; }
  template <class T> __device__ __forceinline__ static int sgn(T d) { return d > 0 ? 1 : (d < 0 ? -1 : 0); }
  // (explicit fused multiply-adds: every product and sum below is an integer smaller than 2^53, so the fused and the unfused forms give the same
  // exact value; the library is built with -ffp-contract=off, which would otherwise keep them apart — 7 and 22 instructions instead of 11 and 31)
  __device__ __forceinline__ static int orient(const P& a, const P& b, const P& c) {
    if constexpr (WIDE) return sgn((a.x - c.x) * (b.y - c.y) - (a.y - c.y) * (b.x - c.x));
    else return sgn(__builtin_fma(a.x - c.x, b.y - c.y, -((a.y - c.y) * (b.x - c.x))));
  }
  __device__ __forceinline__ int orient(int a, int b, int c) const { return orient(pt(a), pt(b), pt(c)); }
  __device__ __forceinline__ static int in_circle(const P& a, const P& b, const P& c, const P& d) {
    const coord_t ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    if constexpr (WIDE) {
      const int32_t al = ax * ax + ay * ay, bl = bx * bx + by * by, cl = cx * cx + cy * cy;
      const int32_t dbc = bx * cy - by * cx, dca = cx * ay - cy * ax, dab = ax * by - ay * bx;
      return sgn((int64_t)al * dbc + (int64_t)bl * dca + (int64_t)cl * dab);
    } else {
      const double al = __builtin_fma(ax, ax, ay * ay), bl = __builtin_fma(bx, bx, by * by), cl = __builtin_fma(cx, cx, cy * cy);
      const double dbc = __builtin_fma(bx, cy, -(by * cx)), dca = __builtin_fma(cx, ay, -(cy * ax)), dab = __builtin_fma(ax, by, -(ay * bx));
      return sgn(__builtin_fma(al, dbc, __builtin_fma(bl, dca, cl * dab)));
    }
  }

  // the 2- and 3-vertex base cases (delaunay.cpp conquer(), triangle.cpp:5964-6060)
  // (the node's vertices are the positions first, first + 1[, first + 2] of the arrangement)
  __device__ void leaf(int first, int n, H& farleft, H& farright, Ctx& c) {
    const int a[3] = {first, first + 1, first + 2};
    if (n == 2) {   // a lone edge: two ghosts glued on all three sides
      farleft = fresh(c);  set_org(farleft, a[0]);  set_dest(farleft, a[1]);
      farright = fresh(c); set_org(farright, a[1]); set_dest(farright, a[0]);
      glue(farleft, farright);
      farleft = cw_edge(farleft); farright = ccw_edge(farright); glue(farleft, farright);
      farleft = cw_edge(farleft); farright = ccw_edge(farright); glue(farleft, farright);
      farleft = cw_edge(farright);
      return;
    }
    H mid = fresh(c), g1 = fresh(c), g2 = fresh(c), g3 = fresh(c);
    const int turn = orient(a[0], a[1], a[2]);
    if (turn == 0) {   // collinear triple: two edges, four ghosts
      set_org(mid, a[0]); set_dest(mid, a[1]);
      set_org(g1, a[1]);  set_dest(g1, a[0]);
      set_org(g2, a[2]);  set_dest(g2, a[1]);
      set_org(g3, a[1]);  set_dest(g3, a[2]);
      glue(mid, g1); glue(g2, g3);
      mid = ccw_edge(mid); g1 = cw_edge(g1); g2 = ccw_edge(g2); g3 = cw_edge(g3);
      glue(mid, g3); glue(g1, g2);
      mid = ccw_edge(mid); g1 = cw_edge(g1); g2 = ccw_edge(g2); g3 = cw_edge(g3);
      glue(mid, g1); glue(g2, g3);
      farleft = g1; farright = g2;
    } else {           // one real triangle ringed by three ghosts
      const int second = turn > 0 ? a[1] : a[2], third = turn > 0 ? a[2] : a[1];
      set_org(mid, a[0]);    set_dest(g1, a[0]);   set_org(g3, a[0]);
      set_dest(mid, second); set_org(g1, second);  set_dest(g2, second);
      set_apex(mid, third);  set_org(g2, third);   set_dest(g3, third);
      glue(mid, g1); mid = ccw_edge(mid);
      glue(mid, g2); mid = ccw_edge(mid);
      glue(mid, g3);
      g1 = cw_edge(g1); g2 = ccw_edge(g2); glue(g1, g2);
      g1 = cw_edge(g1); g3 = cw_edge(g3);  glue(g1, g3);
      g2 = ccw_edge(g2); g3 = cw_edge(g3); glue(g2, g3);
      farleft = g1;
      farright = turn > 0 ? g2 : ccw_edge(farleft);
    }
  }

  // Merge two triangulated halves by walking up the seam between their hulls (delaunay.cpp zip(), triangle.cpp:5638-5947).
  __device__ void zip(H& farleft, H& innerleft, H& innerright, H& farright, int axis, Ctx& c) {
    int il_dest = dest(innerleft), il_apex = apex(innerleft);
    int ir_org = org(innerright), ir_apex = apex(innerright);

    if (axis == 1) {   // horizontal cut: hull handles must point at the extreme-y vertices
      int fl_pt = org(farleft), fl_apex = apex(farleft);
      int fr_pt = dest(farright);
      while (yy(fl_apex) < yy(fl_pt) && --budget > 0) {
        farleft = across(ccw_edge(farleft));
        fl_pt = fl_apex; fl_apex = apex(farleft);
      }
      H probe = across(innerleft); int pv = apex(probe);
      while (yy(pv) > yy(il_dest) && --budget > 0) {
        innerleft = ccw_edge(probe);
        il_apex = il_dest; il_dest = pv;
        probe = across(innerleft); pv = apex(probe);
      }
      while (yy(ir_apex) < yy(ir_org) && --budget > 0) {
        innerright = across(ccw_edge(innerright));
        ir_org = ir_apex; ir_apex = apex(innerright);
      }
      probe = across(farright); pv = apex(probe);
      while (yy(pv) > yy(fr_pt) && --budget > 0) {
        farright = ccw_edge(probe);
        fr_pt = pv;
        probe = across(farright); pv = apex(probe);
      }
    }

    for (bool again = true; again && --budget > 0;) {   // slide down to the lower common tangent
      again = false;
      if (orient(il_dest, il_apex, ir_org) > 0) {
        innerleft = across(cw_edge(innerleft));
        il_dest = il_apex; il_apex = apex(innerleft); again = true;
      }
      if (orient(ir_apex, ir_org, il_dest) > 0) {
        innerright = across(ccw_edge(innerright));
        ir_org = ir_apex; ir_apex = apex(innerright); again = true;
      }
    }

    H lcand = across(innerleft), rcand = across(innerright);
    H base = fresh(c);
    glue(base, innerleft);  base = ccw_edge(base);
    glue(base, innerright); base = ccw_edge(base);
    set_org(base, ir_org); set_dest(base, il_dest);
    if (il_dest == org(farleft)) farleft = ccw_edge(base);
    if (ir_org == dest(farright)) farright = cw_edge(base);

    // The seam walk.  Carried in registers from step to step: the four vertices' coordinates, and — round 5 — each side's FLIP CANDIDATE
    // (the edge across the candidate's far side, its apex w and w's coordinates).  A step changes one side only: the other side's candidate is
    // still what it was (its in-circle test is recomputed, lo_l / lo_r have moved, but nothing is fetched), and the side that advanced
    // fetches its new candidate and that candidate's flip candidate TOGETHER (both hang off the new handle).  A step was nine dependent
    // LDS round trips (three per side and three for the advance), one thread walking alone; now four.
    P lo_l = pt(il_dest), lo_r = pt(ir_org);
    H fe_l, fe_r; int fw_l, fw_r; P fp_l = lo_l, fp_r = lo_r;
    int al = apex(lcand), ar = apex(rcand);
    fe_l = across(cw_edge(lcand)); fe_r = across(ccw_edge(rcand));
    fw_l = apex(fe_l); fw_r = apex(fe_r);
    P up_l = pt(al), up_r = pt(ar);
    if (fw_l >= 0) fp_l = pt(fw_l);
    if (fw_r >= 0) fp_r = pt(fw_r);

    for (;;) {
      if (--budget <= 0) return;
      const bool l_done = orient(up_l, lo_l, lo_r) <= 0;
      const bool r_done = orient(up_r, lo_l, lo_r) <= 0;
      if (l_done && r_done) {
        H cap = fresh(c);
        set_org(cap, lo_l.v); set_dest(cap, lo_r.v);
        glue(cap, base);  cap = ccw_edge(cap);
        glue(cap, rcand); cap = ccw_edge(cap);
        glue(cap, lcand);
        if (axis == 1) {   // back to extreme-x handles
          int fl_pt = org(farleft);
          int fr_pt = dest(farright), fr_apex = apex(farright);
          H probe = across(farleft); int pv = apex(probe);
          while (xx(pv) < xx(fl_pt) && --budget > 0) {
            farleft = cw_edge(probe);
            fl_pt = pv;
            probe = across(farleft); pv = apex(probe);
          }
          while (xx(fr_apex) > xx(fr_pt) && --budget > 0) {
            farright = across(cw_edge(farright));
            fr_pt = fr_apex; fr_apex = apex(farright);
          }
        }
        return;
      }
      if (!l_done) {   // flip away left-hull edges that the new cross edge invalidates
        bool bad = fw_l >= 0 && in_circle(lo_l, lo_r, up_l, fp_l) > 0;
        while (bad && --budget > 0) {
          H e = fe_l;
          e = ccw_edge(e); const H top = across(e);
          e = ccw_edge(e); const H side = across(e);
          glue(e, top);
          glue(lcand, side);
          lcand = ccw_edge(lcand); const H outer = across(lcand);
          e = cw_edge(e);
          glue(e, outer);
          set_org(lcand, lo_l.v); set_dest(lcand, -1); set_apex(lcand, fw_l);
          set_org(e, -1); set_dest(e, up_l.v); set_apex(e, fw_l);
          up_l = fp_l;
          fe_l = side; fw_l = apex(side);
          bad = false;
          if (fw_l >= 0) { fp_l = pt(fw_l); bad = in_circle(lo_l, lo_r, up_l, fp_l) > 0; }
        }
      }
      if (!r_done) {   // same on the right hull, mirrored
        bool bad = fw_r >= 0 && in_circle(lo_l, lo_r, up_r, fp_r) > 0;
        while (bad && --budget > 0) {
          H e = fe_r;
          e = cw_edge(e); const H top = across(e);
          e = cw_edge(e); const H side = across(e);
          glue(e, top);
          glue(rcand, side);
          rcand = cw_edge(rcand); const H outer = across(rcand);
          e = ccw_edge(e);
          glue(e, outer);
          set_org(rcand, -1); set_dest(rcand, lo_r.v); set_apex(rcand, fw_r);
          set_org(e, up_r.v); set_dest(e, -1); set_apex(e, fw_r);
          up_r = fp_r;
          fe_r = side; fw_r = apex(side);
          bad = false;
          if (fw_r >= 0) { fp_r = pt(fw_r); bad = in_circle(lo_l, lo_r, up_r, fp_r) > 0; }
        }
      }
      if (l_done || (!r_done && in_circle(up_l, lo_l, lo_r, up_r) > 0)) {
        glue(base, rcand);
        base = cw_edge(rcand);
        set_dest(base, lo_l.v);
        lo_r = up_r;
        rcand = across(base);
        ar = apex(rcand); fe_r = across(ccw_edge(rcand));       // the new candidate's apex and its flip candidate's edge: independent reads
        fw_r = apex(fe_r);
        up_r = pt(ar);
        if (fw_r >= 0) fp_r = pt(fw_r);
      } else {
        glue(base, lcand);
        base = ccw_edge(lcand);
        set_org(base, lo_r.v);
        lo_l = up_l;
        lcand = across(base);
        al = apex(lcand); fe_l = across(cw_edge(lcand));
        fw_l = apex(fe_l);
        up_l = pt(al);
        if (fw_l >= 0) fp_l = pt(fw_l);
      }
    }
  }
};

// Threads of k_delaunay / k_delaunay_sub.  512, not 1024 (round 6): alone the kernel is slower with fewer (0.75 / 0.79 / 0.89 ms at 1024 / 512 /
// 256: the leaves and the low levels have more nodes than threads), in the pipeline faster — a sixteen-wave workgroup starts only on a CU with
// sixteen free wave slots, which the other slots' eight-wave workgroups keep refilling: 23.25 / 23.75 / 23.87 k pairs/s on the GPU route,
// 6.04 / 6.22 / 6.14 k at 1920x1080 (profiles/r06_dt_threads_ab.txt; k_arrange likewise, kernels.hip).
#ifndef JN_AB_DT_THREADS
#define JN_AB_DT_THREADS 512
#endif
enum { kDtThreads = JN_AB_DT_THREADS, kDtTopThreads = 256, kDtMaxDepth = 16, kDtBytesPerVertex = 32,   // 2 triangles x (4 + 4) halfwords; X, Y, HL, HR ride in the records' spare halfwords (LdsMem)
       kDtMaxPoints = 16384 };                                                                  // arrangement positions are 16-bit, vertex numbers int16 with room to spare

// Sizes and triangle counts per depth: at depth k a node holds f_k = n >> k or f_k + 1 vertices; c[k][b] = count(f_k + b), the triangles a
// subtree of that size creates (a leaf of two vertices 2, of three 4, a merge 2 more than its halves).  K = the depth at which every node
// is a leaf.  Thread 0 fills the tables; the caller synchronises.
DEV void dt_tables(int n, int* s_f, int (*s_c)[2], int* s_K) {
  int K = 0;
  while ((n >> K) > 2) K++;
  for (int k = 0; k <= K + 1; k++) s_f[k] = n >> k;
  for (int k = K + 1; k >= 0; k--)
    for (int b = 0; b < 2; b++) {
      const int m = s_f[k] + b;
      int cnt = 0;
      if (m == 2) cnt = 2; else if (m == 3) cnt = 4;
      else if (m >= 4 && k <= K) { const int h = m >> 1; cnt = s_c[k + 1][h - s_f[k + 1]] + s_c[k + 1][(m - h) - s_f[k + 1]] + 2; }
      s_c[k][b] = cnt;
    }
  *s_K = K;
}
// node (k, j): walk down from the root -> its first position, size and first triangle slot; false when an ancestor is already a leaf
DEV bool dt_node(int n, int k, int j, const int* s_f, const int (*s_c)[2], int& lo, int& size, int& slot) {
  lo = 0; size = n; slot = 0;
  for (int d = 0; d < k; d++) {
    if (size <= 3) return false;
    const int half = size >> 1;
    if ((j >> (k - 1 - d)) & 1) { slot += s_c[d + 1][half - s_f[d + 1]]; lo += half; size -= half; } else size = half;
  }
  return true;
}
// Levels k_deep .. k_top of the tree below node (root_k, root_j), the nodes of a level each on its own thread, a barrier between levels.
// HL / HR: the hull handles a node leaves for its parent, indexed by the node's first position minus hoff (relative handles in LDS).
// Which thread takes which node: neighbouring nodes go to DIFFERENT waves (node i of a round -> wave i % (NT / 64), lane i / (NT / 64)).  The
// merges are data-dependent loops: lanes of one wave that sit in different merges are executed one after the other, so the levels with
// 2 .. 16 nodes — the long merges — took longer in one wave than the root's single merge (measured: 153 us for the two merges below the
// root against 135).
template <class M, int NT, bool WIDE>
DEV bool dt_levels(DT<M, WIDE>& dt, int n, int k_deep, int k_top, int root_k, int root_j, typename M::HullP HL, typename M::HullP HR, int hoff,
                   const int* s_f, const int (*s_c)[2], int tid, long long* dbg_clock, int dbg_side) {
  typedef typename DT<M, WIDE>::H H;
  bool gave_up = false;
  for (int k = k_deep; k >= k_top; k--) {
    const int count = 1 << (k - root_k);
    for (int i0 = 0; i0 < count; i0 += NT) {
      const int i = i0 + (tid & 63) * (NT / 64) + (tid >> 6);
      if (i >= count) continue;
      int lo, size, slot;
      if (!dt_node(n, k, (root_j << (k - root_k)) + i, s_f, s_c, lo, size, slot)) continue;
      H fl, fr;
      if (size <= 3) {
        typename DT<M, WIDE>::Ctx c{slot};
        dt.leaf(lo, size, fl, fr, c);
      } else {
        const int half = size >> 1;
        typename DT<M, WIDE>::Ctx c{slot + s_c[k + 1][half - s_f[k + 1]] + s_c[k + 1][(size - half) - s_f[k + 1]]};
        constexpr int AS = M::kAuxShift;
        fl = (H)(typename M::link_t)HL[(lo - hoff) << AS] + dt.off; fr = (H)(typename M::link_t)HR[(lo + half - hoff) << AS] + dt.off;
        H il = (H)(typename M::link_t)HR[(lo - hoff) << AS] + dt.off, ir = (H)(typename M::link_t)HL[(lo + half - hoff) << AS] + dt.off;
        dt.budget = 16 * size + 256;                                       // (a merge of `size` vertices takes a few steps per seam vertex)
        dt.zip(fl, il, ir, fr, k & 1, c);                                  // the root is cut on axis 0, its children on axis 1, ...
        gave_up |= dt.budget <= 0;
      }
      HL[(lo - hoff) << M::kAuxShift] = (typename M::link_t)(fl - dt.off); HR[(lo - hoff) << M::kAuxShift] = (typename M::link_t)(fr - dt.off);
    }
    if (M::kGlobal) __threadfence();                                       // the next level's threads read what this level's wrote, through memory
    __syncthreads();
    if (dbg_clock && tid == 0) dbg_clock[dbg_side * 32 + k] = wall_clock64();      // (profiling aid: when each level of the tree was done)
  }
  return gave_up;
}
// exclusive scan of one count per thread over the workgroup (NT a multiple of 64, at most 1024): within the wave by shuffles, across the
// waves through LDS (one thread used to add up 1024 entries while 1023 waited).  Returns the thread's offset; *total = the sum.
template <int NT>
DEV int dt_scan(int mine, int tid, int* s_scan, int* total) {
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += up; }
  if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
  __syncthreads();
  if (tid < 64) {
    const int v = tid < NT / 64 ? s_scan[tid] : 0;
    int w = v;
    for (int d = 1; d < NT / 64; d <<= 1) { const int up = __shfl_up(w, d, 64); if (tid >= d) w += up; }
    if (tid < NT / 64) s_scan[64 + tid] = w - v;                           // exclusive wave offsets
    if (tid == NT / 64 - 1) s_scan[128] = w;
  }
  __syncthreads();
  *total = s_scan[128];
  return s_scan[64 + (tid >> 6)] + incl - mine;
}
// FrameInfo and the frame's place in the payload (HostWorker::place): [nsup x (u, v, d)][<= 2 nsup + 8 triangles of the left side][... of the right side]
struct DtPlace { long long base, sup_bytes, side_bytes; };
DEV DtPlace dt_place(FrameInfo* fi, int frame, int side, int n, int nlist, long long payload_stride, int tid) {
  DtPlace p;
  p.base = (long long)frame * payload_stride; p.sup_bytes = (long long)nlist * 12; p.side_bytes = (2ll * nlist + 8) * 12;
  if (side == 0 && tid == 0) {
    fi->nsup = nlist; fi->ok = nlist >= 3 ? 1 : 0;                         // elas.cpp:66-71
    fi->sup_offset = p.base; fi->corner_offset[0] = p.base + p.sup_bytes; fi->corner_offset[1] = p.base + p.sup_bytes + p.side_bytes;
    fi->reserved = n;                                                      // what the list held, clipped or not: the host sizes the next launches by it
  }
  return p;
}

// One workgroup per frame side.  list: (uc, vc, d) int16 triples of the frame's support points in the reference's order; count: how many;
// arr / arr_ok: k_arrange's alternating-cut arrangement of this side's vertices.  Writes FrameInfo (side 0: ok, nsup, the payload
// offsets HostWorker::place() would give the frame at payload_stride * frame; both sides: their ntri), the support points (side 0) and the
// triangles' corner indices into the batch payload.  cap_pts: vertices this launch's LDS holds.
template <bool WIDE>
__global__ void __launch_bounds__(kDtThreads) k_delaunay(const int16_t* __restrict__ list, const int32_t* __restrict__ count, int list_cap, int step,
                                                         const uint16_t* __restrict__ arr, const int32_t* __restrict__ arr_ok, int arr_stride, int cap_pts,
                                                         uint8_t* __restrict__ payload, long long payload_stride, FrameInfo* __restrict__ info,
                                                         int32_t* __restrict__ need_host, long long* __restrict__ dbg_clock) {
  extern __shared__ uint8_t s_dt[];
  __shared__ int s_f[kDtMaxDepth + 2], s_c[kDtMaxDepth + 2][2], s_K;
  __shared__ int s_scan[132];
  const int frame = blockIdx.x, side = blockIdx.y, tid = threadIdx.x;
  const int n = count[frame];
  const int nlist = min(n, list_cap);
  FrameInfo* fi = info + frame;
  const DtPlace pl = dt_place(fi, frame, side, n, nlist, payload_stride, tid);
  if (nlist < 3) { if (tid == 0) fi->ntri[side] = 0; return; }
  const int16_t* t = list + (size_t)frame * list_cap * 3;
  int32_t* sup_out = reinterpret_cast<int32_t*>(payload + pl.base);
  if (side == 0) for (int i = tid; i < nlist; i += kDtThreads) { sup_out[3 * i] = t[3 * i] * step; sup_out[3 * i + 1] = t[3 * i + 1] * step; sup_out[3 * i + 2] = t[3 * i + 2]; }
  if (n > list_cap || n > cap_pts || !arr_ok[frame * 2 + side]) {          // not for this kernel: the host stage takes the batch
    // (the frame is still marked ok and stage B is queued behind this kernel: the OTHER side's triangles index the support points, which is
    // why they were written above whatever happens here — ADVICE r05)
    if (tid == 0) { fi->ntri[side] = 0; atomicOr(&need_host[frame], 1 << side); }
    return;
  }
  // LDS: LINK, VERT [4 T], T = 2 n (count(n) <= 2 n - 2); X, Y, HL, HR in the records' spare halfwords (LdsMem)
  const int np = (n + 3) & ~3, T = 2 * np;
  uint16_t* LINK = reinterpret_cast<uint16_t*>(s_dt); int16_t* VERT = reinterpret_cast<int16_t*>(LINK + 4 * T);
  int16_t* X = reinterpret_cast<int16_t*>(LINK) + 3; int16_t* Y = X + 4;
  uint16_t* HL = reinterpret_cast<uint16_t*>(VERT) + 3; uint16_t* HR = HL + 4;
  const uint16_t* a_in = arr + (size_t)(frame * 2 + side) * arr_stride;
  for (int p = tid; p < n; p += kDtThreads) {                               // vertex p = the p-th of the arrangement
    const int i = a_in[p];
    const int u = t[3 * i] * step, v = t[3 * i + 1] * step, d = t[3 * i + 2];
    X[8 * p] = (int16_t)(side ? u - d : u); Y[8 * p] = (int16_t)v;          // right image: (u - d, v), elas.cpp:466-467
  }
  if (tid == 0) dt_tables(n, s_f, s_c, &s_K);
  __syncthreads();
  DT<LdsMem, WIDE> dt{(lds_ci16*)X, (lds_ci16*)Y, (LdsMem::LinkP)LINK, (LdsMem::VertP)VERT, 0u, 0, 0};
  const int K = s_K;
  long long* clk = frame == 0 ? dbg_clock : nullptr;
  if (clk && tid == 0) clk[side * 32 + 31] = wall_clock64();
  const bool gave_up = dt_levels<LdsMem, kDtThreads, WIDE>(dt, n, K, 0, 0, 0, (LdsMem::HullP)HL, (LdsMem::HullP)HR, 0, s_f, s_c, tid, clk, side);
  if (__syncthreads_or(gave_up)) {                                        // never seen; a structure that does not close must not hang the GPU
    if (tid == 0) { fi->ntri[side] = 0; atomicOr(&need_host[frame], 1 << side); }
    return;
  }
  // output: non-ghost triangles in slot (= creation) order, (org, dest, apex) of edge 0 (delaunay.cpp finish()), positions back to list indices
  const int total = s_c[0][0];
  const int per = (total + kDtThreads - 1) / kDtThreads, t0 = tid * per, t1 = min(t0 + per, total);
  int mine = 0;
  for (int sl = t0; sl < t1; sl++) mine += (VERT[4 * sl] | VERT[4 * sl + 1] | VERT[4 * sl + 2]) >= 0 ? 1 : 0;
  int ntri;
  int out = dt_scan<kDtThreads>(mine, tid, s_scan, &ntri);
  int32_t* tri = reinterpret_cast<int32_t*>(payload + pl.base + pl.sup_bytes + (side ? pl.side_bytes : 0));
  for (int sl = t0; sl < t1; sl++) {
    const int c0 = VERT[4 * sl], c1 = VERT[4 * sl + 1], c2 = VERT[4 * sl + 2];
    if ((c0 | c1 | c2) < 0) continue;
    tri[3 * out] = a_in[c1]; tri[3 * out + 1] = a_in[c2]; tri[3 * out + 2] = a_in[c0];
    out++;
  }
  if (tid == 0) fi->ntri[side] = ntri;
}

// ---- sides beyond one workgroup's LDS: the subtrees below depth C in LDS, the C levels above them in global memory ----
// Global scratch of one frame side, for at most `gcap` vertices: LINK uint32 [8 gcap] | HL, HR uint32 [gcap] each | VERT int16 [8 gcap].
DEV uint32_t* dt_g_link(uint8_t* g, int gcap) { return reinterpret_cast<uint32_t*>(g); }
DEV uint32_t* dt_g_hl(uint8_t* g, int gcap) { return reinterpret_cast<uint32_t*>(g) + 8 * (size_t)gcap; }
DEV uint32_t* dt_g_hr(uint8_t* g, int gcap) { return reinterpret_cast<uint32_t*>(g) + 9 * (size_t)gcap; }
DEV int16_t* dt_g_vert(uint8_t* g, int gcap) { return reinterpret_cast<int16_t*>(reinterpret_cast<uint32_t*>(g) + 10 * (size_t)gcap); }
__host__ __device__ static inline size_t dt_g_bytes(int gcap) { return (size_t)gcap * (10 * 4 + 8 * 2); }

// Subtree (C, blockIdx.z) of a side: its vertices are positions [lo, lo + size) of the arrangement, its triangles slots [slot, slot + count).
template <bool WIDE>
__global__ void __launch_bounds__(kDtThreads) k_delaunay_sub(const int16_t* __restrict__ list, const int32_t* __restrict__ count, int list_cap, int step,
                                                             const uint16_t* __restrict__ arr, const int32_t* __restrict__ arr_ok, int arr_stride, int C, int cap_sub,
                                                             uint8_t* __restrict__ gscratch, int gcap, int32_t* __restrict__ need_host) {
  extern __shared__ uint8_t s_dt[];
  __shared__ int s_f[kDtMaxDepth + 2], s_c[kDtMaxDepth + 2][2], s_K;
  const int frame = blockIdx.x, side = blockIdx.y, jr = blockIdx.z, tid = threadIdx.x;
  const int n = count[frame];
  if (n < 3 || n > list_cap || n > gcap || !arr_ok[frame * 2 + side]) return;       // (k_delaunay_top reports it)
  if (tid == 0) dt_tables(n, s_f, s_c, &s_K);
  __syncthreads();
  int lo, size, slot;
  if (!dt_node(n, C, jr, s_f, s_c, lo, size, slot) || size > cap_sub || (n >> C) <= 3) return;   // (launch_delaunay chose C so that it fits; top checks again)
  const int16_t* t = list + (size_t)frame * list_cap * 3;
  const uint16_t* a_in = arr + (size_t)(frame * 2 + side) * arr_stride;
  const int np = (size + 3) & ~3, T = 2 * np;
  uint16_t* LINK = reinterpret_cast<uint16_t*>(s_dt); int16_t* VERT = reinterpret_cast<int16_t*>(LINK + 4 * T);
  int16_t* X = reinterpret_cast<int16_t*>(LINK) + 3; int16_t* Y = X + 4;
  uint16_t* HL = reinterpret_cast<uint16_t*>(VERT) + 3; uint16_t* HR = HL + 4;
  for (int p = tid; p < size; p += kDtThreads) {
    const int i = a_in[lo + p];
    const int u = t[3 * i] * step, v = t[3 * i + 1] * step, d = t[3 * i + 2];
    X[8 * p] = (int16_t)(side ? u - d : u); Y[8 * p] = (int16_t)v;
  }
  __syncthreads();
  DT<LdsMem, WIDE> dt{(lds_ci16*)X, (lds_ci16*)Y, (LdsMem::LinkP)LINK, (LdsMem::VertP)VERT, 4u * (uint32_t)slot, lo, 0};
  const bool gave_up = dt_levels<LdsMem, kDtThreads, WIDE>(dt, n, s_K, C, C, jr, (LdsMem::HullP)HL, (LdsMem::HullP)HR, lo, s_f, s_c, tid, nullptr, 0);
  if (__syncthreads_or(gave_up)) { if (tid == 0) atomicOr(&need_host[frame], 1 << side); return; }
  // hand the subtree over: records with absolute 32-bit links, the root's hull handles
  uint8_t* g = gscratch + (size_t)(frame * 2 + side) * dt_g_bytes(gcap);
  uint32_t* gl = dt_g_link(g, gcap); int16_t* gv = dt_g_vert(g, gcap);
  const int cnt = s_c[C][size - s_f[C]];
  for (int k = tid; k < 4 * cnt; k += kDtThreads) {
    gl[4 * (size_t)slot + k] = (uint32_t)LINK[k] + 4u * (uint32_t)slot;
    gv[4 * (size_t)slot + k] = VERT[k];
  }
  if (tid == 0) { dt_g_hl(g, gcap)[lo] = (uint32_t)HL[0] + 4u * (uint32_t)slot; dt_g_hr(g, gcap)[lo] = (uint32_t)HR[0] + 4u * (uint32_t)slot; }
}

// The C levels above the subtrees on the global structure, FrameInfo, support points, output.  LDS: the coordinates of ALL the side's vertices.
template <bool WIDE>
__global__ void __launch_bounds__(kDtTopThreads) k_delaunay_top(const int16_t* __restrict__ list, const int32_t* __restrict__ count, int list_cap, int step,
                                                                const uint16_t* __restrict__ arr, const int32_t* __restrict__ arr_ok, int arr_stride, int C, int cap_sub,
                                                                int cap_all, uint8_t* __restrict__ gscratch, int gcap, uint8_t* __restrict__ payload, long long payload_stride,
                                                                FrameInfo* __restrict__ info, int32_t* __restrict__ need_host) {
  extern __shared__ uint8_t s_dt[];
  __shared__ int s_f[kDtMaxDepth + 2], s_c[kDtMaxDepth + 2][2], s_K;
  __shared__ int s_scan[132];
  const int frame = blockIdx.x, side = blockIdx.y, tid = threadIdx.x;
  const int n = count[frame];
  const int nlist = min(n, list_cap);
  FrameInfo* fi = info + frame;
  const DtPlace pl = dt_place(fi, frame, side, n, nlist, payload_stride, tid);
  if (nlist < 3) { if (tid == 0) fi->ntri[side] = 0; return; }
  const int16_t* t = list + (size_t)frame * list_cap * 3;
  int32_t* sup_out = reinterpret_cast<int32_t*>(payload + pl.base);
  if (side == 0) for (int i = tid; i < nlist; i += kDtTopThreads) { sup_out[3 * i] = t[3 * i] * step; sup_out[3 * i + 1] = t[3 * i + 1] * step; sup_out[3 * i + 2] = t[3 * i + 2]; }
  if (tid == 0) dt_tables(max(n, 3), s_f, s_c, &s_K);
  __syncthreads();
  // what the subtrees did not take (they returned without a word), and what they gave up on (need_host is set already)
  const bool tiny = (n >> C) <= 3;                             // a side with next to no vertices under a deep cut: no subtree ran, the whole tree is done here
  bool mine_to_do = !(n > list_cap || n > gcap || n > cap_all || !arr_ok[frame * 2 + side] || (!tiny && s_f[C] + 1 > cap_sub));
  if (mine_to_do && (need_host[frame] >> side) & 1) mine_to_do = false;
  if (!mine_to_do) {
    if (tid == 0) { fi->ntri[side] = 0; atomicOr(&need_host[frame], 1 << side); }
    return;
  }
  const uint16_t* a_in = arr + (size_t)(frame * 2 + side) * arr_stride;
  const int np = (n + 3) & ~3;
  int16_t* X = reinterpret_cast<int16_t*>(s_dt); int16_t* Y = X + np;
  for (int p = tid; p < n; p += kDtTopThreads) {
    const int i = a_in[p];
    const int u = t[3 * i] * step, v = t[3 * i + 1] * step, d = t[3 * i + 2];
    X[p] = (int16_t)(side ? u - d : u); Y[p] = (int16_t)v;
  }
  __syncthreads();
  uint8_t* g = gscratch + (size_t)(frame * 2 + side) * dt_g_bytes(gcap);
  DT<GlobalMem, WIDE> dt{(lds_ci16*)X, (lds_ci16*)Y, (GlobalMem::LinkP)dt_g_link(g, gcap), (GlobalMem::VertP)dt_g_vert(g, gcap), 0u, 0, 0};
  const bool gave_up = dt_levels<GlobalMem, kDtTopThreads, WIDE>(dt, n, tiny ? s_K : C - 1, 0, 0, 0, (GlobalMem::HullP)dt_g_hl(g, gcap), (GlobalMem::HullP)dt_g_hr(g, gcap), 0, s_f, s_c, tid, nullptr, 0);
  if (__syncthreads_or(gave_up)) {
    if (tid == 0) { fi->ntri[side] = 0; atomicOr(&need_host[frame], 1 << side); }
    return;
  }
  const int16_t* gv = dt_g_vert(g, gcap);
  const int total = s_c[0][0];
  const int per = (total + kDtTopThreads - 1) / kDtTopThreads, t0 = tid * per, t1 = min(t0 + per, total);
  int mine = 0;
  for (int sl = t0; sl < t1; sl++) mine += (gv[4 * sl] | gv[4 * sl + 1] | gv[4 * sl + 2]) >= 0 ? 1 : 0;
  int ntri;
  int out = dt_scan<kDtTopThreads>(mine, tid, s_scan, &ntri);
  int32_t* tri = reinterpret_cast<int32_t*>(payload + pl.base + pl.sup_bytes + (side ? pl.side_bytes : 0));
  for (int sl = t0; sl < t1; sl++) {
    const int c0 = gv[4 * sl], c1 = gv[4 * sl + 1], c2 = gv[4 * sl + 2];
    if ((c0 | c1 | c2) < 0) continue;
    tri[3 * out] = a_in[c1]; tri[3 * out + 1] = a_in[c2]; tri[3 * out + 2] = a_in[c0];
    out++;
  }
  if (tid == 0) fi->ntri[side] = ntri;
}

#ifdef JN_HOOKS
__global__ void __launch_bounds__(kDtThreads) k_dt_dummy(int ticks) {
  extern __shared__ uint8_t s_dummy[];
  if (threadIdx.x == 0) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32); }
  __syncthreads();
}
#endif

}  // namespace

int delaunay_gpu_capacity(size_t lds_bytes) { return (int)((lds_bytes > 64 ? lds_bytes - 64 : 0) / kDtBytesPerVertex); }
size_t delaunay_gpu_lds_bytes(int points) { return (size_t)((points + 3) & ~3) * kDtBytesPerVertex + 64; }
int delaunay_gpu_max_points() { return kDtMaxPoints; }
size_t delaunay_gpu_scratch_bytes(int frames, int gcap) { return (size_t)frames * 2 * dt_g_bytes(gcap); }

hipError_t configure_delaunay_kernel() {
#ifdef JN_HOOKS
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_dt_dummy), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
#endif
  const void* ks[] = {reinterpret_cast<const void*>(k_delaunay<false>), reinterpret_cast<const void*>(k_delaunay<true>),
                      reinterpret_cast<const void*>(k_delaunay_sub<false>), reinterpret_cast<const void*>(k_delaunay_sub<true>),
                      reinterpret_cast<const void*>(k_delaunay_top<false>), reinterpret_cast<const void*>(k_delaunay_top<true>)};
  for (const void* k : ks)
    if (const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); e != hipSuccess) return e;
  return hipSuccess;
}

// cap_pts: the most vertices a side of this batch is expected to hold, with a margin (sizes the LDS).  gscratch / gcap (may be null / 0): the
// global scratch of delaunay_gpu_scratch_bytes(n, gcap) bytes that lets sides of up to gcap vertices through, beyond what one workgroup's
// LDS holds; expect_pts: the most vertices a side of the caller's recent batches held (0: unknown) — the whole-side kernel is the faster
// form wherever the sides fit it (profiles/r06_dt_cut_720p_ab.txt), so it is chosen by what the sides ARE, not by the margin on top.
hipError_t launch_delaunay(hipStream_t st, int n, const int16_t* list, const int32_t* count, int list_cap, int step, const uint16_t* arr, const int32_t* arr_ok,
                           int arr_stride, int cap_pts, uint8_t* payload, long long payload_stride, FrameInfo* info, int32_t* need_host, long long* dbg_clock,
                           uint8_t* gscratch, int gcap, int expect_pts, bool wide) {
  static const int whole_env = JN_HOOK_ENV("JN_DT_WHOLE") ? atoi(JN_HOOK_ENV("JN_DT_WHOLE")) : 0;   // (experiment: cut sides the LDS would hold, too)
  const int whole = whole_env ? std::min(whole_env, delaunay_gpu_capacity(152 * 1024)) : delaunay_gpu_capacity(152 * 1024);
  if (JN_HOOK_ENV("JN_DT_FP64")) wide = false;               // (experiment: the FP64 predicates on coordinates they are not exact for)
  if (const hipError_t e = hipMemsetAsync(need_host, 0, sizeof(int32_t) * n, st); e != hipSuccess) return e;
#ifdef JN_HOOKS
  // experiment (what k_delaunay costs the pipeline, and why): a kernel that does nothing for JN_DT_DUMMY_US microseconds behind the real one,
  // JN_DT_DUMMY=1: one wave and no LDS per workgroup (the wait alone), 2: 1024 threads and 152 KB (the wait on a CU nothing else fits on)
  static const int dummy = JN_HOOK_ENV("JN_DT_DUMMY") ? atoi(JN_HOOK_ENV("JN_DT_DUMMY")) : 0;
  static const int dummy_us = JN_HOOK_ENV("JN_DT_DUMMY_US") ? atoi(JN_HOOK_ENV("JN_DT_DUMMY_US")) : 870;
  if (dummy == 1) hipLaunchKernelGGL(k_dt_dummy, dim3(n, 2), dim3(64), 0, st, dummy_us * 100);
  if (dummy == 2) hipLaunchKernelGGL(k_dt_dummy, dim3(n, 2), dim3(kDtThreads), 152 * 1024, st, dummy_us * 100);
  if (dummy == 3) hipLaunchKernelGGL(k_dt_dummy, dim3(n, 2), dim3(64), 152 * 1024, st, dummy_us * 100);
  if (dummy == 4) hipLaunchKernelGGL(k_dt_dummy, dim3(n, 2), dim3(kDtThreads), 0, st, dummy_us * 100);
#endif
  if (cap_pts <= whole || !gscratch || gcap <= whole || (expect_pts > 0 && expect_pts + expect_pts / 16 <= whole)) {
    cap_pts = std::min(cap_pts, whole);
    hipLaunchKernelGGL(wide ? k_delaunay<true> : k_delaunay<false>, dim3(n, 2), dim3(kDtThreads), delaunay_gpu_lds_bytes(cap_pts), st, list, count, list_cap, step, arr, arr_ok,
                       arr_stride, cap_pts, payload, payload_stride, info, need_host, dbg_clock);
    return hipGetLastError();
  }
  // the cut: the smallest depth C whose subtrees (at most (cap_pts >> C) + 1 vertices) fit one workgroup's LDS
  cap_pts = std::min(cap_pts, gcap);
  int C = 1;
  while ((cap_pts >> C) + 1 > whole) C++;
  const int cap_sub = std::min(whole, (cap_pts >> C) + 1);
  hipLaunchKernelGGL(wide ? k_delaunay_sub<true> : k_delaunay_sub<false>, dim3(n, 2, 1 << C), dim3(kDtThreads), delaunay_gpu_lds_bytes(cap_sub), st, list, count, list_cap, step, arr, arr_ok, arr_stride, C, cap_sub,
                     gscratch, gcap, need_host);
  hipLaunchKernelGGL(wide ? k_delaunay_top<true> : k_delaunay_top<false>, dim3(n, 2), dim3(kDtTopThreads), (size_t)((cap_pts + 3) & ~3) * 4 + 64, st, list, count, list_cap, step, arr, arr_ok, arr_stride, C, cap_sub,
                     cap_pts, gscratch, gcap, payload, payload_stride, info, need_host);
  return hipGetLastError();
}

}  // namespace jnav
