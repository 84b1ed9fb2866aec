// k_export.hip -- copy a frame's intermediates out in the reference's layouts
// (locals of RAST:873-964 and saved tensors RAST:998-1019).  Used by the Python
// operator for the backward hook payload (RAST:1128-1140) and by the parity tests.
#include "../../include/gs_rasterizer.h"
#include "gs_common.h"

__global__ void k_export(GsExportArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    switch (a.what) {
    case GS_X_POINT_ID_IN_CAMERA_LIST: if (i < a.M) ((int32_t*)a.dst)[i] = a.ids[i]; break;
    case GS_X_POINT_UV: if (i < a.M) { float4 v = GS_REC(a.PA, i); ((float*)a.dst)[2 * i] = v.x; ((float*)a.dst)[2 * i + 1] = v.y; } break;
    case GS_X_POINT_IN_CAMERA: if (i < a.M) { float4 v = GS_REC(a.PD, i); float* d = (float*)a.dst + 3 * i; d[0] = v.x; d[1] = v.y; d[2] = v.z; } break;
    case GS_X_POINT_UV_CONIC_AND_RESCALE: if (i < a.M) { float4 A = GS_REC(a.PA, i), B = GS_REC(a.PB, i); ((float4*)a.dst)[i] = make_float4(A.z, A.w, B.x, B.y); } break;
    case GS_X_POINT_ALPHA_AFTER_ACTIVATION: if (i < a.M) ((float*)a.dst)[i] = GS_REC(a.PB, i).z; break;
    case GS_X_POINT_COLOR: if (i < a.M) { float4 v = GS_REC(a.PC, i); float* d = (float*)a.dst + 3 * i; d[0] = v.x; d[1] = v.y; d[2] = v.z; } break;
    case GS_X_POINT_RADII: if (i < a.M) ((float*)a.dst)[i] = GS_REC(a.PD, i).w; break;
    case GS_X_NUM_OVERLAP_TILES: if (i < a.M) ((int32_t*)a.dst)[i] = a.ntiles[i]; break;
    case GS_X_ACCUMULATED_NUM_OVERLAP_TILES: if (i < a.M) ((int64_t*)a.dst)[i] = (int64_t)a.offsets[i]; break;
    case GS_X_SORT_KEY:
        if (i < (int64_t)a.K) {
            uint64_t k = a.key64 ? ((const uint64_t*)a.keys_sorted)[i] : (uint64_t)((const uint32_t*)a.keys_sorted)[i];
            int64_t tile = (int64_t)(k >> a.depth_bits);
            int64_t code = (int64_t)(k & ((1ull << a.depth_bits) - 1ull));
            ((int64_t*)a.dst)[i] = code + (tile << 32);                    // RAST:169-170
        }
        break;
    case GS_X_POINT_OFFSET_WITH_SORT_KEY: if (i < (int64_t)a.K) ((int32_t*)a.dst)[i] = a.vals_sorted[i]; break;
    case GS_X_TILE_POINTS_START: if (i < a.T) ((int32_t*)a.dst)[i] = a.tile_start[i]; break;
    case GS_X_TILE_POINTS_END: if (i < a.T) ((int32_t*)a.dst)[i] = a.tile_end[i]; break;
    case GS_X_POINT_DEPTH: if (i < a.M) ((float*)a.dst)[i] = GS_REC(a.PB, i).w; break;
    case GS_X_POINT_IN_CAMERA_MASK: if (i < a.N) ((int8_t*)a.dst)[i] = a.mask[i]; break;
    default: break;
    }
}

void gs_launch_export(const GsExportArgs& a, hipStream_t s)
{
    int64_t n = a.M;
    if (a.what == GS_X_SORT_KEY || a.what == GS_X_POINT_OFFSET_WITH_SORT_KEY) n = a.K;
    else if (a.what == GS_X_TILE_POINTS_START || a.what == GS_X_TILE_POINTS_END) n = a.T;
    else if (a.what == GS_X_POINT_IN_CAMERA_MASK) n = a.N;
    if (n <= 0) return;
    k_export<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a);
}
