/*
 * ORACLE (test infrastructure, not product code) -- polygon -> run-length mask, the part of the
 * COCO mask API that ProposalMaskTarget uses (operator_cxx/proposal_mask_target.cc:19-216 calls
 * rlesInit / rleFrPoly / rleDecode / rlesFree).
 *
 * THIRD PARTY, NOT VENDORED: github.com/RogerChern/cocoapi, common/maskApi.{h,c}
 * (doc/INSTALL.md:90-93; a fork of cocodataset/cocoapi).  This file restates the published
 * algorithm of pycocotools' maskApi.c (rleFrPoly: 5x up-sampled boundary walk, crossings of the
 * x grid, sort, run lengths; rleDecode: column-major fill).  PARITY UNPINNED: no vector produced by
 * the real library exists in the reference; pinned only by properties (tests/test_mask_target.py:
 * axis-aligned rectangles, area of convex polygons, symmetry).
 */
#include "mxshim/coco_api/common/maskApi.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint umax(uint a, uint b) { return a > b ? a : b; }

static int uint_compare(const void *a, const void *b) {
  uint c = *((const uint *)a), d = *((const uint *)b);
  return c > d ? 1 : c < d ? -1 : 0;
}

void rlesInit(RLE **R, siz n) {
  *R = (RLE *)malloc(sizeof(RLE) * (n ? n : 1));
  for (siz i = 0; i < n; i++) { (*R)[i].h = (*R)[i].w = (*R)[i].m = 0; (*R)[i].cnts = NULL; }
}

void rlesFree(RLE **R, siz n) {
  for (siz i = 0; i < n; i++) free((*R)[i].cnts);
  free(*R);
  *R = NULL;
}

void rleFrPoly(RLE *R, const double *xy, siz k, siz h, siz w) {
  /* upsample and get discrete points densely along the entire boundary */
  siz j, m = 0;
  const double scale = 5;
  int *x = (int *)malloc(sizeof(int) * (k + 1)), *y = (int *)malloc(sizeof(int) * (k + 1));
  for (j = 0; j < k; j++) x[j] = (int)(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (j = 0; j < k; j++) y[j] = (int)(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  for (j = 0; j < k; j++) m += umax((uint)abs(x[j] - x[j + 1]), (uint)abs(y[j] - y[j + 1])) + 1;
  int *u = (int *)malloc(sizeof(int) * (m ? m : 1)), *v = (int *)malloc(sizeof(int) * (m ? m : 1));
  m = 0;
  for (j = 0; j < k; j++) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1], dx, dy, t, d;
    dx = abs(xe - xs);
    dy = abs(ys - ye);
    const int flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    const double s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) for (d = 0; d <= dx; d++) {
      t = flip ? dx - d : d; u[m] = t + xs; v[m] = (int)(ys + s * t + .5); m++;
    } else for (d = 0; d <= dy; d++) {
      t = flip ? dy - d : d; v[m] = t + ys; u[m] = (int)(xs + s * t + .5); m++;
    }
  }
  /* get points along the y-boundary and downsample */
  free(x); free(y);
  k = m; m = 0;
  x = (int *)malloc(sizeof(int) * (k ? k : 1));
  y = (int *)malloc(sizeof(int) * (k ? k : 1));
  for (j = 1; j < k; j++) if (u[j] != u[j - 1]) {
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1);
    xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
    yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > h) yd = h;
    yd = ceil(yd);
    x[m] = (int)xd; y[m] = (int)yd; m++;
  }
  /* compute the rle encoding given the y-boundary points */
  k = m;
  uint *a = (uint *)malloc(sizeof(uint) * (k + 1));
  for (j = 0; j < k; j++) a[j] = (uint)(x[j] * (int)(h) + y[j]);
  a[k++] = (uint)(h * w);
  free(u); free(v); free(x); free(y);
  qsort(a, k, sizeof(uint), uint_compare);
  uint p = 0;
  for (j = 0; j < k; j++) { uint t = a[j]; a[j] -= p; p = t; }
  uint *b = (uint *)malloc(sizeof(uint) * k);
  j = m = 0;
  b[m++] = a[j++];
  while (j < k) if (a[j] > 0) b[m++] = a[j++]; else {
    j++;
    if (j < k) b[m - 1] += a[j++];
  }
  R->h = h; R->w = w; R->m = m;
  R->cnts = (uint *)malloc(sizeof(uint) * (m ? m : 1));
  memcpy(R->cnts, b, sizeof(uint) * m);
  free(a); free(b);
}

void rleDecode(const RLE *R, byte *M, siz n) {
  for (siz i = 0; i < n; i++) {
    byte v = 0;
    for (siz j = 0; j < R[i].m; j++) {
      for (siz k = 0; k < R[i].cnts[j]; k++) *(M++) = v;
      v = !v;
    }
  }
}
