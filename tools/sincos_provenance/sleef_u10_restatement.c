// restatement of Sleef 3.x u10 single-precision kernels as torch's CPU build executes them (fast paths), for validation
#include <math.h>
#include <stdint.h>
static inline float fmaf_(float a, float b, float c) { return fmaf(a, b, c); }
typedef struct { float x, y; } f2;
static inline float sin_core(float tx, float ty) {   // sin(t) for the reduced double-float t, as in xsinf_u1 / xcosf_u1
  float s2x = tx * tx;
  float e = fmaf_(tx, tx, -s2x);
  float s2y = fmaf_(tx + tx, ty, e);
  float u = 2.6083159809786593541503e-06f;
  u = fmaf_(u, s2x, -0.0001981069071916863322258f);
  u = fmaf_(u, s2x, 0.00833307858556509017944336f);
  float us = u * s2x;
  const float c = -0.166666597127914428710938f;
  float rx = us + c;
  float ry = (c - rx) + us;
  float px = rx * s2x;
  float pe = fmaf_(rx, s2x, -px);
  float t1 = fmaf_(ry, s2x, pe);
  float py = fmaf_(rx, s2y, t1);
  float xx = px + 1.0f;
  float xy = ((1.0f - xx) + px) + py;
  float m = xy * tx;
  float r0 = fmaf_(ty, xx, m);
  return fmaf_(tx, xx, r0);
}
float sl_sinf(float d) {
  float qf = rintf(d * 0.31830987334251404f);
  int q = (int)qf;
  float u = fmaf_(qf, -3.1414794921875f, d);
  float v5 = qf * -0.0001131594181060791f;
  float v2 = qf * -1.984187258941006e-09f;
  float sx = u + v5;
  float v = sx - u;
  float tx = sx + v2;
  float sy = (u - (sx - v)) + (v5 - v);
  float ty = sy + ((sx - tx) + v2);
  float r = sin_core(tx, ty);
  if (q & 1) r = -r;
  union { float f; uint32_t u; } b; b.f = d;
  if (b.u == 0x80000000u) return d;
  return r;
}
float sl_cosf(float d) {
  float dq = fmaf_(rintf(fmaf_(d, 0.31830987334251404f, -0.5f)), 2.0f, 1.0f);
  int q = (int)dq;
  float y1 = dq * -1.57073974609375f, y2 = dq * -5.657970905303955e-05f, y3 = dq * -9.92093629470503e-10f;
  float s1 = d + y1;
  float v1 = s1 - d;
  float s1y = (d - (s1 - v1)) + (y1 - v1);
  float s2 = s1 + y2;
  float v2 = s2 - s1;
  float s2y = s1y + ((s1 - (s2 - v2)) + (y2 - v2));
  float s3 = s2 + y3;
  float v3 = s3 - s2;
  float s3y = s2y + ((s2 - (s3 - v3)) + (y3 - v3));
  float r = sin_core(s3, s3y);
  if ((q & 2) == 0) r = -r;
  return r;
}
