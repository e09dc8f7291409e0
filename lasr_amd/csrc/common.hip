// common.hip -- error reporting and the optional per-kernel timing of liblasr_hip.so (host code only).
#include <atomic>
#include <mutex>
#include <vector>

#include "host_common.h"

namespace {
thread_local int g_last_hip_error = 0;
// per-kernel timing, scoped to the streams it was switched on for (include/lasr_sr.h): the only state is the set of profiled
// streams and their pending event pairs; n_on lets the common case (nothing profiled) skip the lock
struct ProfRec { hipStream_t st; int id; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<hipStream_t> g_prof_streams;
std::atomic<int> g_prof_n_on{0};
const char* const kKernelNames[K_NUM_KERNELS] = {
    "sr_setup_kernel", "sr_forward_kernel", "sr_backward_kernel",
    "lbs_forward_kernel", "lbs_backward_kernel", "pinhole_forward_kernel", "pinhole_backward_kernel",
    "mask_loss_forward_kernel", "mask_loss_backward_kernel", "flow_loss_stats_kernel", "flow_loss_forward_kernel",
    "flow_loss_backward_kernel", "tex_loss_forward_kernel", "tex_loss_backward_kernel", "loss_finalize_kernel",
    "arap_forward_kernel",
    "arap_backward_kernel", "laplacian_forward_kernel", "laplacian_backward_kernel",
    "flow_reproject_forward_kernel", "flow_reproject_backward_kernel", "quat_forward_kernel", "quat_backward_kernel",
    "skin_forward_kernel", "skin_backward_kernel", "flatten_forward_kernel", "flatten_backward_kernel",
    "face_gather_forward_kernel", "face_gather_backward_kernel", "nearest_point_kernel", "point_mesh_forward_kernel",
    "point_mesh_backward_kernel", "cosdist_forward_kernel", "cosdist_backward_kernel", "load_textures_kernel",
    "geodesic_forward_kernel", "geodesic_backward_kernel", "weighted_means_kernel", "intrinsics_kernel", "bone_fixup_kernel", "chamfer_kernel", "mean_shape_kernel", "obs_pair_kernel", "tail_kernel", "fill_planes_kernel", "gather_rows_kernel", "render_tables_forward_kernel", "render_tables_backward_kernel", "raster_inputs_kernel", "sr_order_kernel",
    "render_tables_flow_kernel", "raster_faces_kernel", "mesh_reg_kernel", "render_tables_fold_kernel", "lbs_backward_fold_kernel", "project_points_kernel", "pose_chain_kernel"};
}  // namespace

int lasr_launch_ok()
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return LASR_E_LAUNCH; }
    return LASR_OK;
}

bool lasr_prof_is_on(hipStream_t st)
{
    if (g_prof_n_on.load(std::memory_order_relaxed) == 0) return false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (hipStream_t s : g_prof_streams) if (s == st) return true;
    return false;
}

void lasr_prof_push(hipStream_t st, int id, hipEvent_t a, hipEvent_t b)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(ProfRec{st, id, a, b});
}

extern "C" int lasr_abi_version(void) { return LASR_ABI_VERSION; }

extern "C" int lasr_last_hip_error(void) { return g_last_hip_error; }

extern "C" const char* lasr_strerror(int code)
{
    switch (code) {
        case LASR_OK: return "ok";
        case LASR_E_BADARG: return "bad argument (null pointer or negative size)";
        case LASR_E_BADMODE: return "mode id out of range";
        case LASR_E_WORKSPACE: return "workspace missing or too small";
        case LASR_E_LAUNCH: return "HIP launch error";
        case LASR_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

extern "C" int lasr_prof_enable(void* hip_stream, int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const hipStream_t st = (hipStream_t)hip_stream;
    size_t i = 0;
    while (i < g_prof_streams.size() && g_prof_streams[i] != st) i++;
    if (on && i == g_prof_streams.size()) g_prof_streams.push_back(st);
    if (!on && i < g_prof_streams.size()) g_prof_streams.erase(g_prof_streams.begin() + i);
    g_prof_n_on.store((int)g_prof_streams.size(), std::memory_order_relaxed);
    return LASR_OK;
}

extern "C" int lasr_prof_kernel_count(void) { return K_NUM_KERNELS; }

extern "C" const char* lasr_prof_kernel_name(int id) { return id >= 0 && id < K_NUM_KERNELS ? kKernelNames[id] : ""; }

// Sums (and clears) the recorded launches of kernel `id`; blocks until they finished.
extern "C" int lasr_prof_collect(void* hip_stream, int id, double* total_ms, long long* launches)
{
    if (!total_ms || !launches) return LASR_E_BADARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0; long long n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : g_prof_recs) {
        if (r.id != id || r.st != (hipStream_t)hip_stream) { keep.push_back(r); continue; }
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; n++; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof_recs.swap(keep);
    *total_ms = tot; *launches = n;
    return LASR_OK;
}
