// PointNet forward / input-gradient (placeholder translation unit: kernels land in pointnet_*.hip).
#include "common.h"

extern "C" int64_t geoa3_pointnet_workspace_bytes(int B, int N, int classes) { return -1; }
extern "C" int geoa3_pointnet_forward(const geoa3_pointnet_weights*, const float*, int, int, float*, void*, void*) {
  return GEOA3_ENOSUPPORT;
}
extern "C" int geoa3_pointnet_backward(const geoa3_pointnet_weights*, const float*, const float*, int, int, float*,
                                       void*, void*) {
  return GEOA3_ENOSUPPORT;
}
