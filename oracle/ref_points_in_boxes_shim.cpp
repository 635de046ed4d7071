// C-ABI caller around the REFERENCE's own CPU source
//   /root/reference/mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cpu.cpp
// which oracle/Makefile compiles from where it lies (never copied here).
// This file only declares the reference function and wraps raw pointers in
// at::Tensor views; it contains none of the reference's arithmetic.
// TEST INFRASTRUCTURE ONLY (see oracle/nesie_oracle.c header).
#include <torch/extension.h>

int points_in_boxes_cpu(at::Tensor boxes_tensor, at::Tensor pts_tensor,
                        at::Tensor pts_indices_tensor);

extern "C" int ref_points_in_boxes_cpu(const float *boxes, int boxes_num,
                                       const float *pts, int pts_num,
                                       int *out /* (boxes_num, pts_num) */) {
  auto fopt = at::TensorOptions().dtype(at::kFloat);
  auto iopt = at::TensorOptions().dtype(at::kInt);
  at::Tensor b = at::from_blob(const_cast<float *>(boxes), {boxes_num, 7}, fopt);
  at::Tensor p = at::from_blob(const_cast<float *>(pts), {pts_num, 3}, fopt);
  at::Tensor o = at::from_blob(out, {boxes_num, pts_num}, iopt);
  return points_in_boxes_cpu(b, p, o);
}
