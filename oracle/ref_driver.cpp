// oracle/ref_driver.cpp — TEST INFRASTRUCTURE, not product code.
//
// Thin C-ABI wrapper around the *real* reference library (libelas as vendored in
// /root/reference/src/elas).  It is compiled by oracle/Makefile together with the reference's own
// .cpp files, straight from where they lie under /root/reference, into oracle/_ref/libelas_ref.so.
// No reference source is copied into this repository; this file only *calls* the reference.
//
// Purpose: (1) pin the CPU restatement in oracle/elas_oracle.cpp stage by stage, (2) generate the
// golden vectors under tests/golden/, (3) serve as `cpu_baseline.kind = "reference"` in bench.py.
//
// The per-stage entry points reach the reference's private members through the usual
// `#define private public` trick, exactly as SURVEY.md Appendix A describes.

#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <emmintrin.h>

#define private public
#include "elas.h"
#undef private
#include "descriptor.h"
#include "triangle.h"

extern "C" {

// Mirror of Elas::parameters (elas.h:60-82) as a plain C struct; same field order.
struct ref_params {
  int32_t disp_min, disp_max;
  float   support_threshold;
  int32_t support_texture, candidate_stepsize, incon_window_size, incon_threshold, incon_min_support;
  int32_t add_corners, grid_size;
  float   beta, gamma, sigma, sradius;
  int32_t match_texture, lr_threshold;
  float   speckle_sim_threshold;
  int32_t speckle_size, ipol_gap_width;
  int32_t filter_median, filter_adaptive_mean, postprocess_only_left, subsampling;
};

static Elas::parameters to_ref(const ref_params* p) {
  Elas::parameters q;
  q.disp_min = p->disp_min; q.disp_max = p->disp_max; q.support_threshold = p->support_threshold;
  q.support_texture = p->support_texture; q.candidate_stepsize = p->candidate_stepsize;
  q.incon_window_size = p->incon_window_size; q.incon_threshold = p->incon_threshold;
  q.incon_min_support = p->incon_min_support; q.add_corners = p->add_corners != 0;
  q.grid_size = p->grid_size; q.beta = p->beta; q.gamma = p->gamma; q.sigma = p->sigma;
  q.sradius = p->sradius; q.match_texture = p->match_texture; q.lr_threshold = p->lr_threshold;
  q.speckle_sim_threshold = p->speckle_sim_threshold; q.speckle_size = p->speckle_size;
  q.ipol_gap_width = p->ipol_gap_width; q.filter_median = p->filter_median != 0;
  q.filter_adaptive_mean = p->filter_adaptive_mean != 0;
  q.postprocess_only_left = p->postprocess_only_left != 0; q.subsampling = p->subsampling != 0;
  return q;
}

void ref_params_default(ref_params* p, int middlebury) {
  Elas::parameters q(middlebury ? Elas::MIDDLEBURY : Elas::ROBOTICS);
  p->disp_min = q.disp_min; p->disp_max = q.disp_max; p->support_threshold = q.support_threshold;
  p->support_texture = q.support_texture; p->candidate_stepsize = q.candidate_stepsize;
  p->incon_window_size = q.incon_window_size; p->incon_threshold = q.incon_threshold;
  p->incon_min_support = q.incon_min_support; p->add_corners = q.add_corners;
  p->grid_size = q.grid_size; p->beta = q.beta; p->gamma = q.gamma; p->sigma = q.sigma;
  p->sradius = q.sradius; p->match_texture = q.match_texture; p->lr_threshold = q.lr_threshold;
  p->speckle_sim_threshold = q.speckle_sim_threshold; p->speckle_size = q.speckle_size;
  p->ipol_gap_width = q.ipol_gap_width; p->filter_median = q.filter_median;
  p->filter_adaptive_mean = q.filter_adaptive_mean;
  p->postprocess_only_left = q.postprocess_only_left; p->subsampling = q.subsampling;
}

// Whole pipeline, exactly the call made at point_cloud.cpp:416-419.
void ref_elas_process(const ref_params* p, const uint8_t* I1, const uint8_t* I2, float* D1,
                      float* D2, int32_t width, int32_t height, int32_t pitch) {
  Elas elas(to_ref(p));
  const int32_t dims[3] = {width, height, pitch};
  elas.process(const_cast<uint8_t*>(I1), const_cast<uint8_t*>(I2), D1, D2, dims);
}

// Opaque per-stage session -------------------------------------------------------------------
struct ref_session {
  Elas*       elas;
  Descriptor* desc1;
  Descriptor* desc2;
  std::vector<Elas::support_pt> sup;
  std::vector<Elas::triangle>   tri1, tri2;
  int32_t*    grid1;
  int32_t*    grid2;
  int32_t     grid_dims[3];
};

ref_session* ref_open(const ref_params* p, const uint8_t* I1_, const uint8_t* I2_, int32_t width,
                      int32_t height, int32_t pitch) {
  ref_session* s = new ref_session();
  s->elas = new Elas(to_ref(p));
  Elas& e = *s->elas;
  e.width = width; e.height = height; e.bpl = width + 15 - (width - 1) % 16;
  e.I1 = (uint8_t*)_mm_malloc(e.bpl * height, 16);
  e.I2 = (uint8_t*)_mm_malloc(e.bpl * height, 16);
  memset(e.I1, 0, e.bpl * height); memset(e.I2, 0, e.bpl * height);
  for (int v = 0; v < height; v++) {
    memcpy(e.I1 + v * e.bpl, I1_ + v * pitch, width);
    memcpy(e.I2 + v * e.bpl, I2_ + v * pitch, width);
  }
  s->desc1 = new Descriptor(e.I1, width, height, e.bpl, e.param.subsampling);
  s->desc2 = new Descriptor(e.I2, width, height, e.bpl, e.param.subsampling);
  s->grid1 = s->grid2 = nullptr;
  return s;
}

void ref_close(ref_session* s) {
  _mm_free(s->elas->I1); _mm_free(s->elas->I2);
  delete s->desc1; delete s->desc2; delete s->elas;
  free(s->grid1); free(s->grid2);
  delete s;
}

// 16 bytes per pixel, [H][W][16]; border (outside u,v in [3,dim-4]) is uninitialised in the reference.
const uint8_t* ref_descriptor(ref_session* s, int right) { return right ? s->desc2->I_desc : s->desc1->I_desc; }

// Sobel responses of one image (filter::sobel3x3 as called by descriptor.cpp:32), [H][bpl].
void ref_sobel(const uint8_t* I, int32_t bpl, int32_t height, uint8_t* du, uint8_t* dv);

// Raw forward/backward support match of one candidate (elas.cpp:269-373).
int32_t ref_match_candidate(ref_session* s, int32_t u, int32_t v, int right) {
  return s->elas->computeMatchingDisparity(u, v, s->desc1->I_desc, s->desc2->I_desc, right != 0);
}

// In-place filters on a caller-owned candidate grid (elas.cpp:153-235).
void ref_remove_inconsistent(ref_session* s, int16_t* D_can, int32_t w, int32_t h) {
  s->elas->removeInconsistentSupportPoints(D_can, w, h);
}
void ref_remove_redundant(ref_session* s, int16_t* D_can, int32_t w, int32_t h, int32_t max_dist,
                          int32_t thresh, int vertical) {
  s->elas->removeRedundantSupportPoints(D_can, w, h, max_dist, thresh, vertical != 0);
}

// Support points (elas.cpp:375-443).  Returns count; fills up to `cap` triples (u,v,d).
int32_t ref_support(ref_session* s, int32_t* uvd, int32_t cap) {
  s->sup = s->elas->computeSupportMatches(s->desc1->I_desc, s->desc2->I_desc);
  int32_t n = (int32_t)s->sup.size();
  for (int32_t i = 0; i < n && i < cap; i++) {
    uvd[3 * i] = s->sup[i].u; uvd[3 * i + 1] = s->sup[i].v; uvd[3 * i + 2] = s->sup[i].d;
  }
  return n;
}

// Replace the session's support points (lets tests feed hand-made point sets to later stages).
void ref_set_support(ref_session* s, const int32_t* uvd, int32_t n) {
  s->sup.clear();
  for (int32_t i = 0; i < n; i++) s->sup.push_back(Elas::support_pt(uvd[3 * i], uvd[3 * i + 1], uvd[3 * i + 2]));
}

// Delaunay + planes (elas.cpp:445-577).  Returns triangle count; per triangle 3 ints + 6 floats.
int32_t ref_triangles(ref_session* s, int right, int32_t* corners, float* planes, int32_t cap) {
  std::vector<Elas::triangle>& t = right ? s->tri2 : s->tri1;
  t = s->elas->computeDelaunayTriangulation(s->sup, right);
  s->elas->computeDisparityPlanes(s->sup, t, right);
  int32_t n = (int32_t)t.size();
  for (int32_t i = 0; i < n && i < cap; i++) {
    corners[3 * i] = t[i].c1; corners[3 * i + 1] = t[i].c2; corners[3 * i + 2] = t[i].c3;
    planes[6 * i] = t[i].t1a; planes[6 * i + 1] = t[i].t1b; planes[6 * i + 2] = t[i].t1c;
    planes[6 * i + 3] = t[i].t2a; planes[6 * i + 4] = t[i].t2b; planes[6 * i + 5] = t[i].t2c;
  }
  return n;
}

// Bare triangulation of an arbitrary float point set through the reference's Triangle build
// (triangle.cpp:8499 with the "zQB" switches of elas.cpp:487).  Returns the triangle count.
int32_t ref_triangulate(const float* xy, int32_t n, int32_t* corners, int32_t cap) {
  struct triangulateio in, out;
  memset(&in, 0, sizeof(in)); memset(&out, 0, sizeof(out));
  in.numberofpoints = n;
  in.pointlist = (float*)malloc(sizeof(float) * 2 * n);
  memcpy(in.pointlist, xy, sizeof(float) * 2 * n);
  char sw[] = "zQB";
  triangulate(sw, &in, &out, NULL);
  int32_t nt = out.numberoftriangles;
  for (int32_t i = 0; i < nt && i < cap; i++) {
    corners[3 * i] = out.trianglelist[3 * i]; corners[3 * i + 1] = out.trianglelist[3 * i + 1];
    corners[3 * i + 2] = out.trianglelist[3 * i + 2];
  }
  free(in.pointlist); free(out.pointlist); free(out.trianglelist);
  return nt;
}

// Grid prior (elas.cpp:579-659).  Returns pointer to [gh][gw][disp_max+2] int32; dims out.
const int32_t* ref_grid(ref_session* s, int right, int32_t* dims3) {
  Elas& e = *s->elas;
  int32_t gw = (int32_t)ceil((float)e.width / (float)e.param.grid_size);
  int32_t gh = (int32_t)ceil((float)e.height / (float)e.param.grid_size);
  s->grid_dims[0] = e.param.disp_max + 2; s->grid_dims[1] = gw; s->grid_dims[2] = gh;
  int32_t*& g = right ? s->grid2 : s->grid1;
  free(g);
  g = (int32_t*)calloc((size_t)(e.param.disp_max + 2) * gh * gw, sizeof(int32_t));
  e.createGrid(s->sup, g, s->grid_dims, right != 0);
  dims3[0] = s->grid_dims[0]; dims3[1] = gw; dims3[2] = gh;
  return g;
}

// Dense matching (elas.cpp:783-907); needs ref_triangles + ref_grid for that side first.
void ref_dense(ref_session* s, int right, float* D) {
  s->elas->computeDisparity(s->sup, right ? s->tri2 : s->tri1, right ? s->grid2 : s->grid1, s->grid_dims,
                            s->desc1->I_desc, s->desc2->I_desc, right != 0, D);
}

void ref_lr_check(ref_session* s, float* D1, float* D2) { s->elas->leftRightConsistencyCheck(D1, D2); }
void ref_speckle(ref_session* s, float* D) { s->elas->removeSmallSegments(D); }
void ref_gap(ref_session* s, float* D) { s->elas->gapInterpolation(D); }
void ref_adaptive_mean(ref_session* s, float* D) { s->elas->adaptiveMean(D); }
void ref_median(ref_session* s, float* D) { s->elas->median(D); }

}  // extern "C"

#include "filter.h"
extern "C" void ref_sobel(const uint8_t* I, int32_t bpl, int32_t height, uint8_t* du, uint8_t* dv) {
  // The reference expects 16-byte aligned buffers (SSE aligned loads).
  uint8_t* a = (uint8_t*)_mm_malloc(bpl * height, 16);
  uint8_t* b = (uint8_t*)_mm_malloc(bpl * height, 16);
  uint8_t* c = (uint8_t*)_mm_malloc(bpl * height, 16);
  memcpy(a, I, bpl * height);
  memset(b, 0, bpl * height); memset(c, 0, bpl * height);
  filter::sobel3x3(a, b, c, bpl, height);
  memcpy(du, b, bpl * height); memcpy(dv, c, bpl * height);
  _mm_free(a); _mm_free(b); _mm_free(c);
}
