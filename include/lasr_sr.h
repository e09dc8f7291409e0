/*
 * lasr_sr.h -- C ABI of the MI355X (gfx950) soft-rasteriser library, liblasr_hip.so.
 *
 * This is the drop-in boundary for the one native extension the LASR hot path
 * needs: `soft_renderer.cuda.soft_rasterize`
 *   /root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp:59-76   forward_soft_rasterize
 *   /root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp:94-114  backward_soft_rasterize
 *   /root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp:135-138 pybind exports
 * Every entry point takes plain device pointers, sizes and a hipStream_t passed
 * as void*; no torch types cross this line.  INTEGRATION.md shows the binding a
 * maintainer of the reference would write against it.
 *
 * Contract (same as the reference, see soft_rasterize.py:41-53,88-89 there):
 *   - all buffers are caller-allocated, dense, fp32, resident on the current HIP device;
 *   - soft_colors comes in pre-filled with the background colour (channels 0-2)
 *     and 1 (channel 3) and is overwritten with the image;
 *   - grad_faces / grad_textures are ACCUMULATED into (the caller zeroes them);
 *   - kernels are enqueued on `stream` (the reference used the legacy default
 *     stream); nothing synchronises the host;
 *   - return value: 0 on success, a negative LASR_E_* code otherwise (the
 *     reference only printf'd launch errors; this library never prints).
 * Mode ids (soft_rasterize.py:22-25): dist {0 hard,1 barycentric,2 euclidean},
 * rgb {0 hard,1 softmax}, alpha {0 hard,1 sum,2 prod}, texture {0 surface,1 vertex}.
 * `dist_eps` is already the logit log(1/dist_eps - 1) (soft_rasterize.py:35).
 */
#ifndef LASR_SR_H_
#define LASR_SR_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASR_OK            0
#define LASR_E_BADARG     (-1)   /* null pointer / negative size                     */
#define LASR_E_BADMODE    (-2)   /* mode id out of range                             */
#define LASR_E_WORKSPACE  (-3)   /* workspace missing or smaller than *_workspace_bytes */
#define LASR_E_LAUNCH     (-4)   /* HIP reported an error at launch (see lasr_last_hip_error) */
#define LASR_E_NODEVICE   (-5)   /* no gfx950 device / code object not loadable       */

/* Version of THIS header.  A caller built against it checks `lasr_abi_version() == LASR_ABI_VERSION` after loading the library
 * (lasr_amd/_lib.py and the INTEGRATION.md stub do); the number moves whenever an exported signature or a size contract changes.
 *   1  rounds 1-3.
 *   2  round 4: lasr_prof_enable(void* stream, int on) / lasr_prof_collect(void* stream, ...) (were process-wide),
 *      lasr_sr_set_forward_math / lasr_sr_set_launch_thresholds removed (per-call lasr_sr_options / flags instead),
 *      lasr_sr_workspace_bytes grows with IS (tile-order table).
 *   3  round 5: lasr_lbs_backward* skip the transform gradients when both pointers are NULL; lasr_skin_weights_backward ignores its
 *      scratch; lasr_render_tables_* run 3 + 2 launches (same signatures and scratch size);
 *      lasr_sr_options gains mixed_min_weight (fifth field); lasr_sr_backward* accept a records-only workspace;
 *      new: lasr_cosdist_multi_*, lasr_raster_faces_*, lasr_mesh_regularisers_*, lasr_project_points_*,
 *      lasr_face_gather_backward_csr, lasr_bone_fixup_pair_*, lasr_step_regularisers_*, lasr_point_mesh_scratch_floats (+ a scratch argument of lasr_point_mesh_forward).
 *      (lasr_lbs_backward_scratch_floats grew about 4x with the 64-vertex chunks of the MFMA backward: re-query it, never cache it per (N, V, K).)
 *   4  round 6: lasr_sr_options.mixed_min_weight (fifth field) became pair_min_tiles -- the one-launch mix of two tile bodies is gone,
 *      launches from that many 8x8 tiles up take the pair-walk forward kernel.  New (no existing signature changes): lasr_pose_chain_*,
 *      lasr_render_tables_forward_imgs; forward flag bits LASR_SR_PAIR_ONE_TEAM / LASR_SR_PAIR_TWO_TEAMS (formerly rejected). */
#define LASR_ABI_VERSION 4
int         lasr_abi_version(void);
const char* lasr_strerror(int code);
int         lasr_last_hip_error(void);      /* hipError_t of the most recent LASR_E_LAUNCH on this thread */

/* Scratch the caller must provide to forward/backward (per-face records + tile
 * bounding boxes; replaces the reference's faces_info tensor, which becomes optional).
 * The backward entry points touch the records and rects only: they accept lasr_sr_workspace_bytes(N, F, T, 0) (the size
 * without the forward's tile-order table; a forward workspace is always large enough). */
size_t lasr_sr_workspace_bytes(int N, int F, int T, int IS);

/*
 * Replaces forward_soft_rasterize (soft_rasterize_cuda.cpp:59-76 -> kernel .cu:674-746).
 *   faces       [N,F,3,3]  in    screen-space face vertices (x, y in NDC, z depth)
 *   textures    [N,F,T,3]  in    T = 3 for vertex colours, R*R for surface texels
 *   faces_info  [N,F,27]   out   OPTIONAL (may be NULL): reference-layout inv|sym|obt|0 record
 *   aggrs_info  [N,2,IS,IS] out  softmax (sum,max) or hard (depth, face index)
 *   soft_colors [N,4,IS,IS] in/out
 */
int lasr_sr_forward(const float* faces, const float* textures, float* faces_info,
                    float* aggrs_info, float* soft_colors,
                    void* workspace, size_t workspace_bytes,
                    int N, int F, int T, int IS,
                    float near, float far, float eps, float sigma_val,
                    int func_id_dist, float dist_eps, float gamma_val,
                    int func_id_rgb, int func_id_alpha, int texture_sample_type,
                    int double_side, void* hip_stream);

/*
 * Replaces backward_soft_rasterize (soft_rasterize_cuda.cpp:94-114 -> kernel .cu:749-813).
 * faces_info is accepted for signature compatibility and ignored (records are
 * rebuilt from `faces` into the workspace; that costs one 1-thread-per-face kernel).
 */
int lasr_sr_backward(const float* faces, const float* textures, const float* soft_colors,
                     const float* faces_info, const float* aggrs_info,
                     float* grad_faces, float* grad_textures, const float* grad_soft_colors,
                     void* workspace, size_t workspace_bytes,
                     int N, int F, int T, int IS,
                     float near, float far, float eps, float sigma_val,
                     int func_id_dist, float dist_eps, float gamma_val,
                     int func_id_rgb, int func_id_alpha, int texture_sample_type,
                     int double_side, void* hip_stream);

/*
 * Same two calls with near/far read from device memory (`near_far_dev` -> {near, far}, two floats).  LASR derives
 * the clipping planes from the projected vertices on every iteration (nnutils/mesh_net.py:304-311) and the
 * reference turns those 0-dim device tensors into Python floats at each extension call, i.e. one device->host
 * synchronisation per call; these variants keep the values on the device.
 */
int lasr_sr_forward_dev(const float* faces, const float* textures, float* faces_info,
                        float* aggrs_info, float* soft_colors,
                        void* workspace, size_t workspace_bytes,
                        int N, int F, int T, int IS,
                        const float* near_far_dev, float eps, float sigma_val,
                        int func_id_dist, float dist_eps, float gamma_val,
                        int func_id_rgb, int func_id_alpha, int texture_sample_type,
                        int double_side, void* hip_stream);
int lasr_sr_backward_dev(const float* faces, const float* textures, const float* soft_colors,
                         const float* faces_info, const float* aggrs_info,
                         float* grad_faces, float* grad_textures, const float* grad_soft_colors,
                         void* workspace, size_t workspace_bytes,
                         int N, int F, int T, int IS,
                         const float* near_far_dev, float eps, float sigma_val,
                         int func_id_dist, float dist_eps, float gamma_val,
                         int func_id_rgb, int func_id_alpha, int texture_sample_type,
                         int double_side, void* hip_stream);

/*
 * Multi-attribute rasterisation.  LASR's flow renders rasterise the SAME geometry twice
 * in one call, once per 3-channel vertex attribute (camera-space positions of frame t and t', nnutils/mesh_net.py:85-87).
 * These entry points interpolate and depth-blend `channels` (3, 6 or 9) per-vertex attributes in one pass: the per-channel
 * arithmetic is that of the 3-channel kernels, so channels [0,3), [3,6) and [6,9) equal separate renders and the face
 * gradient equals the sum of theirs.  Only LASR's mode combination (2,1,2,1, double sided) is accepted for 6 / 9 channels.
 *   textures [N,F,3,channels]   soft_colors / grad_soft_colors [N,channels+1,IS,IS] (alpha is the LAST plane)
 *   grad_textures [N,F,3,channels]   near_far_dev: optional {near, far} on the device (else the two floats are used)
 */
int lasr_sr_forward_attr(const float* faces, const float* textures, float* aggrs_info, float* soft_colors,
                         void* workspace, size_t workspace_bytes, int N, int F, int channels, int IS,
                         float near, float far, const float* near_far_dev, float eps, float sigma_val,
                         int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                         int texture_sample_type, int double_side, void* hip_stream);
int lasr_sr_backward_attr(const float* faces, const float* textures, const float* soft_colors,
                          const float* aggrs_info, float* grad_faces, float* grad_textures,
                          const float* grad_soft_colors, void* workspace, size_t workspace_bytes, int N, int F,
                          int channels, int IS, float near, float far, const float* near_far_dev, float eps,
                          float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                          int func_id_alpha, int texture_sample_type, int double_side, void* hip_stream);

/*
 * Supersets of the entry points above with every option as an argument (no reference counterpart).  channels = 3, 6 or 9 (6 and 9 as
 * for the *_attr variants); near_far_dev: NULL or a device pointer to {near, far} that overrides near / far.
 * forward flags  : 0, or LASR_SR_RELAXED_MATH -- LASR's mode combination only: the point-to-face distance and the
 *                  `dis >= threshold` decision stay bit-faithful, the sigmoid, alpha product, clip / normalise, depth and
 *                  softmax weights use fp32 v_rcp / v_exp arithmetic (~15 % faster forward, image within ~3e-5 of the default
 *                  arithmetic; the north-star bar is 1e-4).  LASR_SR_DEFAULT_FLAGS (-1) is accepted and means 0.
 *                  LASR_SR_SEGMENTED -- LASR's mode combination, launches small enough for the several-waves-per-tile kernels
 *                  (lasr_sr_options): a tile's face list is split into index-ordered segments folded by 4 or 8 waves in
 *                  parallel and the partial (alpha product, depth-softmax) states are merged at the end instead of one wave
 *                  applying every fragment in face order.  Exact arithmetic is unchanged (the aggregates are symmetric in the
 *                  fragments), the rounding sequence is not: image within ~1e-6 of the default path.  Larger launches and other
 *                  modes ignore the flag.  The backward pass reads whatever aggregates the forward wrote: no flag needed.
 * backward flags : LASR_SR_RECORDS_VALID -- the caller vouches that `workspace` still holds the per-face records the forward
 *                  pass of the SAME faces / N / F / IS / sigma_val / dist_eps left there (nothing else was run on that
 *                  workspace in between); the backward then skips its own setup launch.  Without the flag every backward
 *                  call rebuilds the records, so the plain entry points stay safe for callers that share one workspace.
 *                  LASR_SR_GRADS_OVERWRITE -- vertex textures only: every element of grad_faces / grad_textures is written (each
 *                  face is owned by one wavefront), so the caller may pass uninitialised buffers instead of zeroed ones (the
 *                  reference accumulates with atomics into zeroed tensors, soft_rasterize.py:88-89); ignored for surface textures.
 */
#define LASR_SR_DEFAULT_FLAGS (-1)   /* forward only: the default arithmetic (same as 0) */
#define LASR_SR_RELAXED_MATH  1
#define LASR_SR_SEGMENTED     2   /* forward, LASR's modes, small launches: see below */
#define LASR_SR_RECORDS_VALID 4
#define LASR_SR_GRADS_OVERWRITE 8   /* backward, vertex textures: grad_faces / grad_textures need not be zeroed by the caller */
#define LASR_SR_PAIR_ONE_TEAM  16  /* forward, pair-walk kernel: one team of four waves per 16x16 tile whatever the launch size (round 6) */
#define LASR_SR_PAIR_TWO_TEAMS 32  /* forward, pair-walk kernel: two teams per tile (the default up to 8192 tiles); image within 1e-6 */
int lasr_sr_forward_ex(const float* faces, const float* textures, float* faces_info, float* aggrs_info, float* soft_colors,
                       void* workspace, size_t workspace_bytes, int N, int F, int T, int channels, int IS, float near,
                       float far, const float* near_far_dev, float eps, float sigma_val, int func_id_dist, float dist_eps,
                       float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                       int flags, void* hip_stream);
/* lasr_sr_forward_ex with the background colour as an argument (`background`: HOST array of `channels` floats): soft_colors
 * need not be pre-filled -- every element is written, colour planes and alpha -- which saves the caller's fill pass
 * (soft_rasterize.py:50-53) and the kernel's read of it. */
int lasr_sr_forward_bg(const float* faces, const float* textures, float* faces_info, float* aggrs_info, float* soft_colors,
                       void* workspace, size_t workspace_bytes, int N, int F, int T, int channels, int IS, float near,
                       float far, const float* near_far_dev, float eps, float sigma_val, int func_id_dist, float dist_eps,
                       float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                       const float* background, int flags, void* hip_stream);
int lasr_sr_backward_ex(const float* faces, const float* textures, const float* soft_colors, const float* aggrs_info,
                        float* grad_faces, float* grad_textures, const float* grad_soft_colors, void* workspace,
                        size_t workspace_bytes, int N, int F, int T, int channels, int IS, float near, float far,
                        const float* near_far_dev, float eps, float sigma_val, int func_id_dist, float dist_eps,
                        float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                        int flags, void* hip_stream);

/*
 * float64 tensors (the reference dispatches its kernels on the tensor type, AT_DISPATCH_FLOATING_TYPES at
 * soft_rasterize_cuda_kernel.cu:701,716,780; the scalar arguments stay float there as here).  Same call shape as
 * lasr_sr_forward / lasr_sr_backward, three colour channels, every mode combination; `faces_info` [N,F,27] doubles is written
 * by the forward pass when given (the reference's layout) and then serves the backward pass; when NULL both passes build the
 * per-face data in `workspace` (lasr_sr_workspace_bytes_f64).  soft_colors holds the background on entry, gradients accumulate
 * into zeroed buffers, as for the reference.  A brute-force path in the reference's operation order (csrc/sr_fp64.hip): LASR
 * never renders in double, this exists so that the operator accepts what the reference's accepts.
 */
size_t lasr_sr_workspace_bytes_f64(int N, int F);
int lasr_sr_forward_f64(const double* faces, const double* textures, double* faces_info, double* aggrs_info,
                        double* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T, int IS,
                        float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                        int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side, void* hip_stream);
int lasr_sr_backward_f64(const double* faces, const double* textures, const double* soft_colors, const double* faces_info,
                         const double* aggrs_info, double* grad_faces, double* grad_textures, const double* grad_soft_colors,
                         void* workspace, size_t workspace_bytes, int N, int F, int T, int IS, float near, float far, float eps,
                         float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                         int texture_sample_type, int double_side, void* hip_stream);

/*
 * Per-call launch options (no reference counterpart).  The library keeps NO mutable process state: what used to be
 * process-wide setters in rounds 1-3 (forward arithmetic, kernel-choice thresholds) is an argument now, so two callers in one
 * process with different settings cannot race.
 * Which forward kernel a launch of LASR's mode combination takes is decided by its size in 8x8-pixel tiles (frames x tiles
 * per frame); the output is bit-identical whichever kernel runs (tests/test_forward_kernel_choice_gpu.py runs every one on
 * the same inputs).  Up to coop8_max_tiles: eight waves share a tile; up to coop_max_tiles: four waves (csrc/sr_forward_coop.h:
 * latency designs for launches that cannot fill the chip); up to choose_max_tiles: a one-wave kernel estimates the BUSY tiles
 * from the meshes' pixel bounding boxes on the device and picks four waves per tile (estimate at most coop_max_tiles) or one
 * wave per tile -- both are launched, the one not chosen returns at once; above: one wave per tile.  A NEGATIVE field takes the
 * built-in default: 2200 / 14336 / 49152 (six and nine channels: 5/8 of the first two and no device-decided range; measured
 * on an MI355X, csrc/sr_raster.hip), or the value of LASR_SR_COOP8_MAX_TILES / LASR_SR_COOP_MAX_TILES /
 * LASR_SR_CHOOSE_MAX_TILES read ONCE when the library is loaded.
 * order_max_tiles: launches of LASR's mode combination with five frames or more (tile total a multiple of 8) and at most this many tiles issue their
 * tiles HEAVIEST FIRST -- two small kernels (sr_tile_weight_kernel, one workgroup per image; sr_order_kernel, one per XCD)
 * count, from the setup kernel's pixel rects, the faces that touch each 8x8 tile of this batch and write the block -> tile
 * table the forward kernels then follow, instead of the fixed centre-out order.  Which block renders a tile does not change
 * the tile's arithmetic: output bit-identical; forward + order kernels are 6-25 % faster at 6-128 frames, even at 256; below five frames the two order launches (12 us) cost more than they gain
 * (profiles/r04_tile_order_ab.txt).  For such launches four waves share a tile up to 4/7 of coop_max_tiles, the device-decided
 * range ends at 7/16 of choose_max_tiles, and the device decides on the count of non-empty tiles (at most 3/8 of
 * coop_max_tiles: four waves) instead of the bounding-box estimate.  An XCD sorts all of its ceil(N / 8) images together while
 * their face records stay within 8 MB, twice its L2 (2420 faces: up to 136 frames); larger launches (multiples of 8 frames) sort
 * and issue them four images at a time (all 32 images of an XCD at once, 256 frames: record fetch 172 MB -> 1.18 GB per launch).
 * Default (negative): no limit but the kernels' capacity (images up to 1016 pixels a side); LASR_SR_ORDER_MAX_TILES at load time;
 * 0 switches it off.
 */
typedef struct lasr_sr_options {
    long long coop8_max_tiles;
    long long coop_max_tiles;
    long long choose_max_tiles;
    long long order_max_tiles;
    long long pair_min_tiles;     /* LASR's mode combination: launches of at least this many 8x8-pixel tiles (frames x tiles per
                                     frame) take the pair-walk kernel (sr_forward_pairs.h: every lane walks the (pixel, face) pairs
                                     of its own pixel, records staged in LDS); 0 = every launch, a huge value = never.  Its image
                                     agrees with the other kernels' to ~1e-6 (another accumulation order per pixel), not bit for
                                     bit.  Negative: the built-in default (LASR_SR_PAIR_MIN_TILES at load time) */
} lasr_sr_options;
/* lasr_sr_forward_bg with options: `background` may be NULL (then soft_colors holds the pre-filled background, as for
 * lasr_sr_forward_ex), `options` may be NULL (all defaults). */
int lasr_sr_forward_opt(const float* faces, const float* textures, float* faces_info, float* aggrs_info, float* soft_colors,
                        void* workspace, size_t workspace_bytes, int N, int F, int T, int channels, int IS, float near,
                        float far, const float* near_far_dev, float eps, float sigma_val, int func_id_dist, float dist_eps,
                        float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                        const float* background, int flags, const lasr_sr_options* options, void* hip_stream);

/*
 * Optional per-kernel timing for benchmarks (no reference counterpart: the reference has no profiling hooks, SURVEY.md
 * section 5), scoped to a STREAM: while enabled for `hip_stream`, every kernel this library launches on that stream is
 * bracketed by hipEvents; launches on other streams (another trainer in the same process) are neither timed nor slowed.
 * lasr_prof_collect blocks until the recorded launches of that stream finished and returns their summed duration and count,
 * then forgets them.
 */
int         lasr_prof_enable(void* hip_stream, int on);
int         lasr_prof_kernel_count(void);
const char* lasr_prof_kernel_name(int kernel_id);
int         lasr_prof_collect(void* hip_stream, int kernel_id, double* total_ms, long long* launches);

/*
 * lasr_sr_peek_choice (test hook, synchronises the stream): the device-side choice word of the LAST forward call on
 * `workspace`, meaningful only if that call's size was in the device-decided range (lasr_sr_options).  Launches in the fixed
 * tile order: sr_choose_kernel's decision, 0 one wave per tile, 1 four waves per tile.  Launches in their own tile order
 * (lasr_sr_options order_max_tiles): the number of non-empty tiles sr_order_kernel counted, which the forward kernels compare with
 * 3/8 of coop_max_tiles themselves.
 */
int lasr_sr_peek_choice(const void* workspace, int N, int F, int* choice, void* hip_stream);


/*
 * Test hook (no reference counterpart): adds to *mismatches the number of pairs for which the library's
 * exact division-by-reciprocal (sr_device.h) differs bitwise from the IEEE quotient a[i] / b[i].
 */
int lasr_selftest_div(const float* a, const float* b, int* mismatches, int n, void* hip_stream);
/* Same for the shared-divisor division of the clip/normalise step: a [n,3] with 0 <= a <= 1, b [n] in [1e-5, 3];
 * adds the number of quotients (3 per row) that differ bitwise from a[i,k] / b[i]. */
int lasr_selftest_div3(const float* a, const float* b, int* mismatches, int n, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LASR_SR_H_ */
