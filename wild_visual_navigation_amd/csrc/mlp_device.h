// Device helpers shared by the traversability-MLP training kernels (mlp.hip: one kernel per stage; mlp_train.hip: the
// four-launch step): the confidence statistic and the confidence of a reconstruction loss.
// Reference: wild_visual_navigation/utils/confidence_generator.py:78-82, 182-193.
#pragma once
#include "common.h"

struct ConfStats { float mean, std; };
__device__ inline ConfStats conf_stats(const double* st) {
  const double n = st[0];
  const double mean = st[1] / n;
  const double var = (st[2] - st[1] * st[1] / n) / (n - 1.0);  // unbiased (torch.std); NaN for n < 2
  ConfStats c;
  c.mean = (float)mean;
  c.std = (float)sqrt(var > 0.0 || !(var == var) ? var : 0.0);
  return c;
}
// confidence_generator.py:182-193
__device__ inline float confidence_of(float x, float mean, float std, float f) {
  const float shifted = mean + std * f;
  float lo = shifted - std;
  lo = (lo > 0.f || isnan(lo)) ? lo : 0.f;  // python max(lo, 0): NaN stays NaN
  const float hi = shifted + std;
  float xc = fminf(fmaxf(x, lo), hi);
  if (isnan(lo) || isnan(hi)) xc = NAN;
  return 1.f - (xc - lo) / (hi - lo);
}

