// host_common.h -- host-side helpers shared by the .hip translation units of liblasr_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lasr_sr.h"

// kernel ids for lasr_prof_* (names in common.hip, same order)
enum LasrKernelId {
    K_SR_SETUP = 0, K_SR_FORWARD, K_SR_BACKWARD,
    K_LBS_FORWARD, K_LBS_BACKWARD, K_PINHOLE_FORWARD, K_PINHOLE_BACKWARD,
    K_MASK_LOSS_FORWARD, K_MASK_LOSS_BACKWARD, K_FLOW_LOSS_STATS, K_FLOW_LOSS_FORWARD, K_FLOW_LOSS_BACKWARD,
    K_TEX_LOSS_FORWARD, K_TEX_LOSS_BACKWARD, K_LOSS_FINALIZE, K_ARAP_FORWARD, K_ARAP_BACKWARD, K_LAP_FORWARD, K_LAP_BACKWARD,
    K_FLOW_REPROJECT_FORWARD, K_FLOW_REPROJECT_BACKWARD, K_QUAT_FORWARD, K_QUAT_BACKWARD, K_SKIN_FORWARD, K_SKIN_BACKWARD,
    K_FLATTEN_FORWARD, K_FLATTEN_BACKWARD, K_FACE_GATHER_FORWARD, K_FACE_GATHER_BACKWARD,
    K_NEAREST_POINT, K_POINT_MESH_FORWARD, K_POINT_MESH_BACKWARD, K_COSDIST_FORWARD, K_COSDIST_BACKWARD,
    K_LOAD_TEXTURES, K_GEODESIC_FORWARD, K_GEODESIC_BACKWARD, K_WEIGHTED_MEANS, K_INTRINSICS, K_BONE_FIXUP, K_CHAMFER, K_MEAN_SHAPE, K_OBS_PAIR, K_TAIL, K_FILL_PLANES, K_GATHER_ROWS, K_RENDER_TABLES_FORWARD, K_RENDER_TABLES_BACKWARD, K_RASTER_INPUTS, K_SR_ORDER,
    K_RENDER_TABLES_FLOW, K_RASTER_FACES, K_MESH_REG, K_RENDER_TABLES_FOLD, K_LBS_BACKWARD_FOLD, K_PROJECT_POINTS, K_POSE_CHAIN,
    K_NUM_KERNELS
};

int lasr_launch_ok();                                   // hipGetLastError -> LASR_OK / LASR_E_LAUNCH (records the code)
bool lasr_prof_is_on(hipStream_t st);
void lasr_prof_push(hipStream_t st, int id, hipEvent_t a, hipEvent_t b);

// Brackets one kernel launch with hipEvents on its stream when profiling is enabled.
struct ProfScope {
    hipStream_t st; int id; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(int id_, hipStream_t st_) : st(st_), id(id_), on(lasr_prof_is_on(st_))
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(b, st); lasr_prof_push(st, id, a, b); }
    }
};

#define launch_ok lasr_launch_ok
