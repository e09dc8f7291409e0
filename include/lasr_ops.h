/*
 * lasr_ops.h -- C ABI of the geometry / loss kernels that sit around the rasteriser on the LASR hot
 * path (same library, liblasr_hip.so; same conventions as lasr_sr.h: device pointers, fp32, sizes,
 * hipStream_t as void*, 0 / negative LASR_E_* return codes, nothing printed, nothing synchronised).
 *
 * The reference implements all of these as chains of eager PyTorch ops (no native interface to bind
 * to); each entry point names the Python function it replaces.  Paths are relative to /root/reference/.
 */
#ifndef LASR_OPS_H_
#define LASR_OPS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Linear-blend skinning, nnutils/geom_utils.py:45-71 (obj_to_cam):
 *   vs[n,v]  = sum_{k=1..K-1} skin[n,k-1,v] * (verts[n,v] @ R[n*K+k] + T[n*K+k])      (K > 1; else vs = verts)
 *   out[n,v] = tocam ? vs[n,v] @ R[n*K] + T[n*K] : vs[n,v]
 * verts [N,V,3], Rmat [N*K,3,3] (row-vector convention, bone-major inside a mesh), Tmat [N*K,3],
 * skin [N,K-1,V] (may be NULL when K == 1), out [N,V,3].
 * Forward: the blend contraction skin^T[V,K-1] x RT[K-1,12] runs on the matrix cores
 * (v_mfma_f32_16x16x4_f32, exact fp32).  Backward overwrites (does not accumulate) all four gradients;
 * it is deterministic (no atomics).  Any gradient pointer may be NULL to skip it.
 */
int lasr_lbs_forward(const float* verts, const float* Rmat, const float* Tmat, const float* skin, float* out,
                     int N, int V, int K, int tocam, void* hip_stream);
/* Both results of one blend: out_cam (tocam = 1) and out_blend (the vertices before the body transform, tocam = 0) -- what
 * LASR.forward needs as verts_cam and deform_v (nnutils/mesh_net.py:291,298; two obj_to_cam calls in the reference). */
int lasr_lbs_forward_both(const float* verts, const float* Rmat, const float* Tmat, const float* skin, float* out_cam,
                          float* out_blend, int N, int V, int K, void* hip_stream);
size_t lasr_lbs_backward_scratch_floats(int N, int V, int K);     /* chunk partials of the transform gradients */
/* Backward.  The three contractions (blended transform, g_skin, the transposed g_RT) run on v_mfma_f32_16x16x4_f32 (K <= 65;
 * more bones: the VALU kernel of ABI version 1); a second small launch folds the chunk partials of grad_Rmat / grad_Tmat.  With
 * grad_Rmat == grad_Tmat == NULL that contraction and the fold are skipped (scratch may then be NULL). */
int lasr_lbs_backward(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                      const float* grad_out, float* grad_verts, float* grad_Rmat, float* grad_Tmat,
                      float* grad_skin, float* scratch, int N, int V, int K, int tocam, void* hip_stream);
int lasr_lbs_backward_both(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                           const float* grad_out_cam, const float* grad_out_blend, float* grad_verts, float* grad_Rmat,
                           float* grad_Tmat, float* grad_skin, float* scratch, int N, int V, int K, void* hip_stream);

/*
 * Joint centres and control points into the image: the two obj_to_cam calls with an identity skin and the pinhole_cam of
 * nnutils/mesh_net.py:285-288, :302 in one launch each way.  rest_ts / ctl_ts [H, K-1, 3]; Rmat [M*K,3,3], Tmat [M*K,3] with
 * M = images x H (hypothesis fastest); pp [M/H, 2] per image, fl [M].  Point j of hypothesis h rides on part bone (j mod (K-1)) + 1
 * and then the body transform (bone 0) of every m = img * H + h:  proj [M, 2(K-1), 4] = (pp + xy * fl / z, z, 1), joints first.
 * The transforms and intrinsics are constants (the reference detaches them); backward: grad_rest / grad_ctl [H, K-1, 3], summed
 * over the images in image order (overwritten).  Same association of the sums as lasr_lbs_forward + lasr_pinhole_forward.
 */
int lasr_project_points_forward(const float* rest_ts, const float* ctl_ts, const float* Rmat, const float* Tmat, const float* pp,
                                const float* fl, float* proj, int M, int H, int K, void* hip_stream);
int lasr_project_points_backward(const float* rest_ts, const float* ctl_ts, const float* Rmat, const float* Tmat, const float* fl,
                                 const float* grad_proj, float* grad_rest, float* grad_ctl, int M, int H, int K, void* hip_stream);

/*
 * The pose chain of LASR.forward in ONE launch each way (round 6; nnutils/mesh_net.py:204-217 intrinsics bookkeeping, :232 quaternion ->
 * matrix, :259-283 bone fix-up with the rotation distance of :514-516, :285-288 + :302 joint / control-point projection): the
 * phases of lasr_intrinsics_*, lasr_quat_to_rotmat_*, lasr_bone_fixup_pair_* and lasr_project_points_* run by one workgroup, with
 * the `.repeat(1, H, 1, 1)` of trans / depth (:237-238) inside.  B image pairs, H hypotheses, K bones, M = 2B*H:
 *   in   cams [2B, cam_stride] (column 0 = crop scale), pp [2B,2], scale [2B,H], depth [2B,K], ppoint [2B,2], quat4 [M*K,4] (x,y,z,w),
 *        trans [2B*K,2], rest_ts / ctl_ts [H,K-1,3] (NULL when K == 1)
 *   out  scale_out [2B,H], depth_out [2B,K], ppoint_out [2B,2], trans_rep [M*K,2], depth_rep [M*K], rmat [M*K,3,3], tmat [M*K,3],
 *        pair_angle [M*K/2] (NULL: not wanted), proj [M, 2(K-1), 4] (K > 1)
 * Backward: gradients of the outputs (any of grad_scale_out / grad_ppoint_out / grad_trans_rep / grad_depth_rep / grad_pair_angle /
 * grad_proj may be NULL = zero) -> gradients of scale, depth, ppoint, quat4, trans, rest_ts, ctl_ts (overwritten); the
 * projection treats transforms and intrinsics as constants like lasr_project_points_backward; scratch = M*K*3 floats.
 */
int lasr_pose_chain_forward(const float* cams, int cam_stride, const float* pp, const float* scale, const float* depth,
                            const float* ppoint, const float* quat4, const float* trans, const float* rest_ts, const float* ctl_ts,
                            float* scale_out, float* depth_out, float* ppoint_out, float* trans_rep, float* depth_rep, float* rmat,
                            float* tmat, float* pair_angle, float* proj, int B, int H, int K, float half_size, void* hip_stream);
int lasr_pose_chain_backward(const float* cams, int cam_stride, const float* quat4, const float* rest_ts, const float* ctl_ts,
                             const float* rmat, const float* tmat, const float* scale_out, const float* grad_scale_out,
                             const float* grad_ppoint_out, const float* grad_trans_rep, const float* grad_depth_rep,
                             const float* grad_rmat, const float* grad_tmat, const float* grad_pair_angle, const float* grad_proj,
                             float* grad_scale, float* grad_depth, float* grad_ppoint, float* grad_quat4, float* grad_trans,
                             float* grad_rest, float* grad_ctl, float* scratch, int B, int H, int K, void* hip_stream);

/*
 * Pinhole projection, nnutils/geom_utils.py:27-34 (pinhole_cam), with pp / fl already expanded per mesh:
 *   out.x = pp[n,0] + x * fl[n] / z ;  out.y = pp[n,1] + y * fl[n] / z ;  out.z = z ; out.w = w
 * verts/out [N,V,4], pp [N,2], fl [N].  Backward overwrites grad_verts [N,V,4], grad_pp [N,2], grad_fl [N].
 */
int lasr_pinhole_forward(const float* verts, const float* pp, const float* fl, float* out, int N, int V,
                         void* hip_stream);
int lasr_pinhole_backward(const float* verts, const float* pp, const float* fl, const float* grad_out,
                          float* grad_verts, float* grad_pp, float* grad_fl, int N, int V, void* hip_stream);

/*
 * The three loss tables below split every (image, hypothesis) row into chunks so that small tables still fill
 * the chip, and fold the chunk partials in a fixed order.  `scratch` (lasr_loss_scratch_floats(I,H,P) floats,
 * caller-allocated) carries the partials and the per-row counts from a forward call to the matching backward call.
 */
size_t lasr_loss_scratch_floats(int I, int H, int P);

/*
 * Silhouette loss table, nnutils/mesh_net.py:374-388:
 *   loss[i,j] = 0.5 * mean_{p : occ[i,p] != 0} (mask_pred[i,j,p] - masks[i,p])^2
 * mask_pred [I,H,P], masks [I,P], occ [I,P] -> loss [I,H].  (Empty selection -> NaN like torch's mean of
 * an empty tensor.)  Backward: grad_pred [I,H,P] = grad_loss[i,j] * (pred - mask) / count on selected pixels.
 */
int lasr_mask_loss_forward(const float* mask_pred, const float* masks, const float* occ, float* loss,
                           float* scratch, int I, int H, int P, void* hip_stream);
int lasr_mask_loss_backward(const float* mask_pred, const float* masks, const float* occ, const float* grad_loss,
                            const float* scratch, float* grad_pred, int I, int H, int P, void* hip_stream);

/*
 * Optical-flow loss table, nnutils/mesh_net.py:393-413:
 *   sel[i,j,p] = !bg[i,j,p] && occ[i,p] != 0 && masks[i,p] > 0
 *   w[i,p]     = sigmoid(-occ[i,p]) / mean_{(j,p) in sel[i]} sigmoid(-occ[i,p])
 *   loss[i,j]  = 0.5 * mean_{p in sel[i,j]} ||flow_rd[i,j,p,:] - flow_obs[i,:,p]||_2 * w[i,p]     (0 if sel[i,j] empty)
 * flow_rd [I,H,P,2], flow_obs: channel planes of P floats, images obs_image_stride floats apart (the first two
 * channels of the [I,3,P] observation), bg [I,H,P] uint8,
 * occ/masks [I,P] -> loss [I,H], and the weighted error map flow_rd_map [I,H,P] the trainer logs.
 * Backward: gradient w.r.t. flow_rd only (the weights are data).
 */
int lasr_flow_loss_forward(const float* flow_rd, const float* flow_obs, const unsigned char* bg, const float* occ,
                           const float* masks, float* loss, float* flow_rd_map, float* scratch,
                           int I, int H, int P, int obs_image_stride, void* hip_stream);
int lasr_flow_loss_backward(const float* flow_rd, const float* flow_obs, const unsigned char* bg, const float* occ,
                            const float* masks, const float* scratch, const float* grad_loss, float* grad_flow_rd,
                            int I, int H, int P, int obs_image_stride, void* hip_stream);
/* The same, also writing sel as vis_mask [I,H,P] uint8 (the `vis_mask` LASR.forward returns, nnutils/mesh_net.py:405). */
int lasr_flow_loss_forward_vis(const float* flow_rd, const float* flow_obs, const unsigned char* bg, const float* occ,
                               const float* masks, float* loss, float* flow_rd_map, unsigned char* vis_mask, float* scratch,
                               int I, int H, int P, int obs_image_stride, void* hip_stream);

/*
 * L1 texture loss table, nnutils/mesh_net.py:425-441 (without the perceptual term):
 *   loss[i,j] = 2*wt * ( mean_{occ[i]!=0} mean_c |img_obs[i,c,p] - rnd[i,j,c,p]*fg[i,j,p]|
 *                      + mean_{occ[i]!=0} mean_c |img_white[i,c,p] - rnd[i,j,c,p]| )
 * img_obs/img_white [I,3,P], rnd [I,H,3,P], fg [I,H,P], occ [I,P] -> loss [I,H].
 * Backward: grad_rnd [I,H,3,P] and grad_fg [I,H,P] (overwritten).
 */
int lasr_tex_loss_forward(const float* img_obs, const float* img_white, const float* rnd, const float* fg,
                          const float* occ, float* loss, float* scratch, float wt, int I, int H, int P,
                          void* hip_stream);
int lasr_tex_loss_backward(const float* img_obs, const float* img_white, const float* rnd, const float* fg,
                           const float* occ, const float* grad_loss, const float* scratch, float* grad_rnd,
                           float* grad_fg, float wt, int I, int H, int P, void* hip_stream);

/*
 * Mesh regularisers on sparse adjacency instead of dense [V,V] operators.
 *
 * ARAP, nnutils/loss_utils.py:46-64:  loss[n] = mean over directed edges (v,u), u in nbr(v), of
 *   | ||x[n,u]-x[n,v]||^2 - ||dx[n,u]-dx[n,v]||^2 |     (the reference builds six dense [N,V,V] tensors for this).
 * Laplacian, third_party/ext_nnutils/loss_utils.py:34-65:  loss[n] = sum_v || x_v - mean_{u in nbr(v)} x_u ||^2 ;
 *   vertices without neighbours contribute 0 (the reference leaves their row at zero, :50-51).
 * Both take the mesh adjacency as CSR (row_ptr [V+1], col [nnz], unique neighbours, symmetric), x/dx [N,V,3]
 * -> loss [N].  Backward entry points overwrite the gradients and are deterministic (vertex-centric gathers).
 */
int lasr_arap_forward(const float* dx, const float* x, const int* row_ptr, const int* col, float* loss,
                      int N, int V, void* hip_stream);
int lasr_arap_backward(const float* dx, const float* x, const int* row_ptr, const int* col, const float* grad_loss,
                       float* grad_dx, float* grad_x, int N, int V, void* hip_stream);
int lasr_laplacian_forward(const float* x, const int* row_ptr, const int* col, float* loss, int N, int V,
                           void* hip_stream);
int lasr_laplacian_backward(const float* x, const int* row_ptr, const int* col, const float* grad_loss,
                            float* grad_x, float* scratch_lx /*[N,V,3]*/, int N, int V, void* hip_stream);

/*
 * The three shape regularisers of a step in ONE launch each way (nnutils/mesh_net.py:449-459 Laplacian + flatten on the mean shape
 * x [N,V,3]; :494-497 ARAP between the two frames' deformed shapes arap_dx / arap_x [NA,V,3]); the same arithmetic and summation
 * orders as lasr_laplacian_* / lasr_flatten_* / lasr_arap_*, results bit-identical to calling those one by one:
 *   lap_loss [N], flat_loss [N], arap_loss [NA];  lap_coords [N,V,3] = the Laplacian coordinates, kept by the caller for the backward.
 * Index buffers as for the separate entry points (CSR adjacency of the Laplacian and of ARAP, quads [E,4], incidence CSR of the
 * flatten backward).  Backward: grad_x [N,V,3] = Laplacian part + flatten part (overwritten), grad_arap_dx / grad_arap_x [NA,V,3]
 * (either may be NULL).  N or NA may be 0.
 */
int lasr_mesh_regularisers_forward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                   const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                   float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, int N, int NA, int V,
                                   int E, void* hip_stream);
int lasr_mesh_regularisers_backward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                    const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                    const int* inc_ptr, const int* inc, const float* lap_coords, const float* grad_lap,
                                    const float* grad_flat, const float* grad_arap, float* grad_x, float* grad_arap_dx,
                                    float* grad_arap_x, int N, int NA, int V, int E, void* hip_stream);
/* The same two launches with a fourth criterion riding along: the symmetric squared Chamfer distance of NC pairs of small point sets
 * (the bones' control points against their mirror images, nnutils/mesh_net.py:500-503) -- cham_a [NC,P,3], cham_b [NC,Q,3] ->
 * cham_loss [NC], nn_ab [NC,P] / nn_ba [NC,Q] (int32, kept by the caller for the backward); the values and gradients of
 * lasr_chamfer_forward / _backward, bit for bit, without their two launches.  NC = 0: exactly lasr_mesh_regularisers_*. */
int lasr_step_regularisers_forward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                   const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                   float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, int N, int NA, int V,
                                   int E, const float* cham_a, const float* cham_b, float* cham_loss, int* nn_ab, int* nn_ba, int NC,
                                   int P, int Q, void* hip_stream);
int lasr_step_regularisers_backward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                    const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                    const int* inc_ptr, const int* inc, const float* lap_coords, const float* grad_lap,
                                    const float* grad_flat, const float* grad_arap, float* grad_x, float* grad_arap_dx,
                                    float* grad_arap_x, int N, int NA, int V, int E, const float* cham_a, const float* cham_b,
                                    const int* nn_ab, const int* nn_ba, const float* grad_cham, float* grad_cham_a,
                                    float* grad_cham_b, int NC, int P, int Q, void* hip_stream);

/*
 * Flow reprojection, nnutils/mesh_net.py:93-104 (the tail of render_flow_soft_2 after the render).
 * px [N,7,P] = the 6-attribute render (lasr_sr_forward_attr): planes 0-2 camera-space position of frame t seen at each
 * pixel, planes 3-5 of frame t', plane 6 alpha.  bgmask = (px[2] < 1e-9) | (px[5] < 1e-9); background pixels take the
 * point (10,10,10) (:93-95); flow[n,p] = proj(p1; pp1[n], fl1[n]) - proj(p0; pp0[n], fl0[n]) with
 * proj(q; pp, fl) = pp + (q.xy * fl) / q.z (:98-101).  pp0/pp1 [N,2], fl0/fl1 [N] -> flow [N,P,2], bgmask [N,P] (0/1 bytes).
 * Backward: the frame-t projection and all background pixels are detached (:102-103), so the gradient reaches
 * px planes 3-5 (grad_px [N,7,P] is overwritten whole, zeros elsewhere), pp1 [N,2] and fl1 [N].
 * scratch: lasr_flow_reproject_scratch_floats(N, P) floats.
 */
size_t lasr_flow_reproject_scratch_floats(int N, int P);
int lasr_flow_reproject_forward(const float* px, const float* pp0, const float* pp1, const float* fl0, const float* fl1,
                                float* flow, unsigned char* bgmask, int N, int P, void* hip_stream);
int lasr_flow_reproject_backward(const float* px, const float* fl1, const float* grad_flow, float* grad_px,
                                 float* grad_pp1, float* grad_fl1, float* scratch, int N, int P, void* hip_stream);

/*
 * The same on six position planes that sit inside a wider render (LASR.forward renders texture + both position triples in
 * one 9-attribute pass, [N,10,P]): pos6 points at the first position plane of image 0, consecutive images are batch_stride
 * floats apart (>= 6 P), the six planes of an image are contiguous.  grad_pos6 is a dense [N,6,P] tensor.
 */
int lasr_flow_reproject_planes_forward(const float* pos6, long long batch_stride, const float* pp0, const float* pp1,
                                       const float* fl0, const float* fl1, float* flow, unsigned char* bgmask, int N, int P,
                                       void* hip_stream);
int lasr_flow_reproject_planes_backward(const float* pos6, long long batch_stride, const float* fl1, const float* grad_flow,
                                        float* grad_pos6, float* grad_pp1, float* grad_fl1, float* scratch, int N, int P,
                                        void* hip_stream);

/*
 * Unit quaternion -> rotation matrix, kornia 0.5.3 `quaternion_to_rotation_matrix` semantics (not vendored in the
 * reference; call sites nnutils/mesh_net.py:232,250,265 and third_party/ext_nnutils/net_blocks.py:359): coefficient
 * order (x,y,z,w); the quaternion is normalised first (q / max(|q|, 1e-12));
 *   R = [[1-2(yy+zz), 2(xy-zw), 2(xz+yw)], [2(xy+zw), 1-2(xx+zz), 2(yz-xw)], [2(xz-yw), 2(yz+xw), 1-2(xx+yy)]].
 * quat [M,4] -> rotmat [M,9] row-major.  Backward: grad_rotmat [M,9] -> grad_quat [M,4] (overwritten).
 */
int lasr_quat_to_rotmat_forward(const float* quat, float* rotmat, int M, void* hip_stream);
int lasr_quat_to_rotmat_backward(const float* quat, const float* grad_rotmat, float* grad_quat, int M, void* hip_stream);

/*
 * GMM skinning weights, nnutils/mesh_net.py:264-271:
 *   skin[h,k,v] = softmax over k of  -10 * sum_d exp(log_ctl[h,k,d]) * ((ctl_ts[h,k] - verts[h,v]) * R(ctl_rs[h,k]))_d^2
 * ctl_ts, log_ctl [H*J,3], ctl_rs [H*J,4] (x,y,z,w, normalised inside), verts [H,V,3] (frame-0 mean shape, a constant:
 * :266 detaches it) -> skin [H,J,V].  J <= 64.
 * Backward: grad_skin [H,J,V] -> grad_ts [H*J,3], grad_rs [H*J,4], grad_log_ctl [H*J,3] (overwritten);
 * scratch: H*V floats.
 */
int lasr_skin_weights_forward(const float* ctl_ts, const float* ctl_rs, const float* log_ctl, const float* verts,
                              float* skin, int H, int J, int V, void* hip_stream);
int lasr_skin_weights_backward(const float* ctl_ts, const float* ctl_rs, const float* log_ctl, const float* verts,
                               const float* skin, const float* grad_skin, float* grad_ts, float* grad_rs,
                               float* grad_log_ctl, float* scratch, int H, int J, int V, void* hip_stream);

/*
 * Flatten loss, third_party/ext_nnutils/loss_utils.py:110-152: loss[n] = sum over the listed interior edges of
 * (cos + 1)^2 where cos is the cosine between the parts of (v2 - v0) and (v3 - v0) orthogonal to the edge (v1 - v0),
 * every eps = 1e-6 as in the reference (:120-147).  quads [E,4] int32 = (v0,v1,v2,v3) per edge (built on the host from
 * the faces, :73-108), x [N,V,3] -> loss [N].
 * Backward: per-edge gradients into scratch (N*E*12 floats), then a vertex-centric gather through the incidence lists
 * inc_ptr [V+1], inc [4E] (entry = edge*4 + slot, ascending) -> grad_x [N,V,3] (overwritten, deterministic).
 */
int lasr_flatten_forward(const float* x, const int* quads, float* loss, int N, int V, int E, void* hip_stream);
int lasr_flatten_backward(const float* x, const int* quads, const int* inc_ptr, const int* inc, const float* grad_loss,
                          float* grad_x, float* scratch, int N, int V, int E, void* hip_stream);

/*
 * Per-face gather of per-vertex attributes, third_party/softras/soft_renderer/functional/face_vertices.py:4-22:
 *   out[n,f,c,:] = attr[n, faces[n,f,c], :]      attr [N,V,C], faces [N,F,3] int64 (torch's index type) -> out [N,F,3,C].
 * Indices must lie in [0,V) (the reference asserts nothing either; out-of-range is undefined behaviour).
 * Backward: grad_out [N,F,3,C] -> grad_attr [N,V,C], overwritten; the sum over a vertex's corners runs in ascending
 * corner order (deterministic; the reference's autograd uses atomic index_add_).
 */
int lasr_face_gather_forward(const float* attr, const long long* faces, float* out, int N, int V, int F, int C,
                             void* hip_stream);
int lasr_face_gather_backward(const float* grad_out, const long long* faces, float* grad_attr, int N, int V, int F, int C,
                              void* hip_stream);
/* The same backward over a CSR incidence structure the caller built once for the connectivity (the layout of
 * lasr_raster_faces_backward: inc_ptr int32 [N or 1, V+1], inc int32 [N or 1, 3F] = corner ids 3 f + c grouped by vertex, ascending
 * inside a vertex; inc_shared = 1 when all meshes share one).  Same summation order, hence the same bits, without the scan of the
 * face tensor: for callers whose connectivity outlives a call (LASR's does; the Python operator caches the structure per face
 * tensor). */
int lasr_face_gather_backward_csr(const float* grad_out, const int* inc_ptr, const int* inc, int inc_shared, float* grad_attr,
                                  int N, int V, int F, int C, void* hip_stream);

/*
 * Brute-force nearest neighbour between two small point sets (the idx1/dist1 outputs of third_party/chamfer3D/chamfer3D.cu
 * as used at nnutils/mesh_net.py:477, and the inner minimum of pytorch3d's chamfer_distance, :503):
 * a [N,P,3], b [N,Q,3] -> d2 [N,P] squared distance to, and idx [N,P] (int32) index of, the nearest b point
 * (lowest index on ties).  No backward entry point: the caller differentiates |a - b[idx]|^2 with the indices fixed.
 */
int lasr_nearest_point(const float* a, const float* b, float* d2, int* idx, int N, int P, int Q, void* hip_stream);

/*
 * Point <-> triangle-mesh distance, pytorch3d.loss.point_mesh_face_distance as used at nnutils/mesh_net.py:470-471
 * (pytorch3d 0.4.0 is not vendored: semantics restated, parity unpinned):
 *   dmin_point[n,p] = min_f d2(points[n,p], tri[n,f]),  dmin_face[n,f] = min_p d2(points[n,p], tri[n,f])
 * with the arg-min indices (int32, lowest index on ties); the caller forms mean_n(mean_p dmin_point + mean_f dmin_face).
 * verts [N,V,3], faces [F,3] int64 shared by the batch, points [N,P,3].
 * Backward: given d loss / d dmin_point = grad_point_term and d loss / d dmin_face = grad_face_term (uniform weights,
 * as the means produce), writes grad_tri [N,F,3,3] (per face corner; reduce to vertices with lasr_face_gather_backward)
 * and grad_points [N,P,3].  d d2/d p = 2 (p - q), d d2/d corner_i = -2 w_i (p - q) for the closest point q = sum w_i corner_i.
 */
size_t lasr_point_mesh_scratch_floats(int N, int F, int P);            /* per-chunk minima of the two-launch forward */
int lasr_point_mesh_forward(const float* verts, const long long* faces, const float* points, float* dmin_point,
                            int* arg_point, float* dmin_face, int* arg_face, float* scratch, int N, int V, int F, int P,
                            void* hip_stream);
int lasr_point_mesh_backward(const float* verts, const long long* faces, const float* points, const int* arg_point,
                             const int* arg_face, float grad_point_term, float grad_face_term, float* grad_tri,
                             float* grad_points, int N, int V, int F, int P, void* hip_stream);

/*
 * Perceptual-distance reduction, third_party/PerceptualSimilarity/util/util.py:71-83 (normalize_tensor, cos_sim) and
 * models/networks_basic.py:51-57 (1 - cos_sim per feature layer), the reduction behind `ptex_loss.forward_pair` at
 * nnutils/mesh_net.py:442:   dist[n] = 1 - mean_p sum_c a_hat[c,p] * b_hat[c,p],   x_hat = x / (sqrt(sum_c x_c^2) + 1e-10).
 * feat_obs [N/rep, C, P] (features of the observed images, each shared by `rep` consecutive rendered images -- the
 * reference feeds rep identical copies through the network), feat_rnd [N, C, P] -> dist [N].  scratch:
 * lasr_cosdist_scratch_floats(N, P).  Backward: gradient w.r.t. feat_rnd only (grad_rnd [N,C,P], overwritten); the
 * observed side is data.  Where a rendered feature vector is exactly zero the reference's autograd yields NaN
 * (sqrt'(0)); here the ill-defined term is dropped (finite gradient).
 */
size_t lasr_cosdist_scratch_floats(int N, int P);
int lasr_cosdist_forward(const float* feat_obs, const float* feat_rnd, float* dist, float* scratch, int N, int C, int P,
                         int rep, void* hip_stream);
int lasr_cosdist_backward(const float* feat_obs, const float* feat_rnd, const float* grad_dist, float* grad_rnd, int N,
                          int C, int P, int rep, void* hip_stream);

/*
 * The same reduction over ALL feature layers of the perceptual network in one launch each way (the sum over layers of
 * models/networks_basic.py:51-64): dist[n] = sum_l (1 - mean_p cos_l[n,p]), layers added in list order, each layer's value
 * bit-identical to lasr_cosdist_forward's.  feat_obs / feat_rnd / grad_rnd: HOST arrays of n_layers device pointers
 * ([N/rep, C_l, P_l] / [N, C_l, P_l]); C, P: host arrays; n_layers <= LASR_COSDIST_MAX_LAYERS.
 * scratch: lasr_cosdist_multi_scratch_floats floats (tile partials; a second one-wave-per-image launch folds them).
 * Backward: every grad_rnd[l] overwritten.
 */
#define LASR_COSDIST_MAX_LAYERS 8
size_t lasr_cosdist_multi_scratch_floats(const int* P, int n_layers, int N);
int lasr_cosdist_multi_forward(const float* const* feat_obs, const float* const* feat_rnd, const int* C, const int* P,
                               int n_layers, float* dist, float* scratch, int N, int rep, void* hip_stream);
int lasr_cosdist_multi_backward(const float* const* feat_obs, const float* const* feat_rnd, const int* C, const int* P,
                                int n_layers, const float* grad_dist, float* const* grad_rnd, int N, int rep, void* hip_stream);

/*
 * Texture atlas -> per-face surface textures, replaces `soft_renderer.cuda.load_textures`
 * (third_party/softras/soft_renderer/cuda/load_textures_cuda.cpp:10-28, kernel load_textures_cuda_kernel.cu:8-66; called
 * from functional/load_obj.py when scripts/render_syn.py:71 loads its textured mesh).  image [H,W,3] (row 0 first, the caller
 * flips as load_obj.py does), faces_uv [F,3,2] in [0,1], is_update [F] or NULL (NULL = all), textures [F,R*R,3] (only the
 * faces with is_update != 0 are written).  Texel (w_x, w_y): barycentric ((w_x + 1/3)/R, (w_y + 1/3)/R, rest) for
 * w_x + w_y < R, else the mirrored upper-triangle position; bilinear sample at uv * (size - 1).
 */
int lasr_load_textures(const float* image, const float* faces_uv, const int* is_update, float* textures, int F, int R, int H,
                       int W, void* hip_stream);

/*
 * ---- small-tensor glue of LASR.forward as single kernels (lasr_amd/csrc/glue.hip) -----------------------------------
 *
 * Rotation distance, third_party/ext_utils/util_rot.py:27-37 (called at nnutils/mesh_net.py:508 / :516): m1, m2 [n,3,3]
 * row-major -> angle [n] = acos((trace(m1 m2^T) - 1) / 2); where |cos| >= 1 the angle is 0 / pi with zero gradient (the
 * reference's acos(min(cos, 1)) back-propagates NaN there and its trainer skips the step).
 */
int lasr_geodesic_forward(const float* m1, const float* m2, float* angle, int n, void* hip_stream);
int lasr_geodesic_backward(const float* m1, const float* m2, const float* grad_angle, float* grad_m1, float* grad_m2, int n,
                           void* hip_stream);

/*
 * total = sum_t weights[t] * mean(terms[t]) -- the reference's chain `total_loss += w * x.mean()` at
 * nnutils/mesh_net.py:374-530.  terms: HOST array of n_terms device pointers (each a contiguous fp32 tensor of numels[t]
 * elements), weights / groups: host arrays; out [n_groups + 1] (device): out[g] = sum of the weighted means of group g (the
 * per-loss scalars LASR logs: mask_loss, flow_rd_loss, ...), out[n_groups] = total, both accumulated in term order.
 * Backward: coef[t] = grad_total * weights[t] / numels[t] (the gradient of every element of term t).
 */
#define LASR_MEANS_MAX_TERMS 24
int lasr_weighted_means_forward(const float* const* terms, const int* numels, const float* weights, const int* groups,
                                int n_terms, int n_groups, float* out, void* hip_stream);
int lasr_weighted_means_backward(const int* numels, const float* weights, int n_terms, const float* grad_total, float* coef,
                                 void* hip_stream);

/*
 * Intrinsics bookkeeping of the image pair, nnutils/mesh_net.py:204-217.  cams [2B, cam_stride] (column 0 = crop scale; rows
 * 0..B-1 frames t, B..2B-1 frames t'), pp [2B,2], predicted scale [2B,H], depth [2B,K], ppoint [2B,2], half_size = img_size/2:
 *   scale_out = cams0 * scale;  depth_out = depth with column 0 multiplied by cams0;
 *   ppoint_out[:B] = ppoint[:B];  ppoint_out[B+i] = (ppoint[i] + cams0_i pp_i / half + 1) * (cams0_{B+i} / cams0_i)
 *                                                   - cams0_{B+i} pp_{B+i} / half - 1
 * Backward: gradients w.r.t. scale, depth, ppoint (rows B.. of ppoint receive 0: the reference discards that prediction).
 */
int lasr_intrinsics_forward(const float* cams, int cam_stride, const float* pp, const float* scale, const float* depth,
                            const float* ppoint, float* scale_out, float* depth_out, float* ppoint_out, int B, int H, int K,
                            float half_size, void* hip_stream);
int lasr_intrinsics_backward(const float* cams, int cam_stride, const float* grad_scale_out, const float* grad_depth_out,
                             const float* grad_ppoint_out, float* grad_scale, float* grad_depth, float* grad_ppoint, int B, int H,
                             int K, void* hip_stream);

/*
 * Bone-transform fix-up, nnutils/mesh_net.py:259-283.  quat [M,K,9]: the 3x3 Q the pose head
 * predicts per (image-hypothesis m = image * H + h, bone k); trans [M*K,2], depth [M*K]; rest_ts [H,K-1,3] joint centres.
 *   root k = 0:  rmat = Q^T, tmat = (trans, depth)
 *   bone k >= 1: rmat = Q,   tmat = -Q^T c + (trans, depth) + c,  c = rest_ts[h, k-1]      (rotation about the joint)
 * rmat [M*K,3,3], tmat [M*K,3].  Backward: gradients of quat, trans, depth and rest_ts (summed over the images, in image
 * order: deterministic).  rest_ts / grad_rest may be NULL when K == 1.
 */
int lasr_bone_fixup_forward(const float* quat, const float* trans, const float* depth, const float* rest_ts, float* rmat,
                            float* tmat, int M, int H, int K, void* hip_stream);
int lasr_bone_fixup_backward(const float* quat, const float* rest_ts, const float* grad_rmat, const float* grad_tmat,
                             float* grad_quat, float* grad_trans, float* grad_depth, float* grad_rest, int M, int H, int K,
                             void* hip_stream);
/* The same with the rotation distance between the two frames of every pair riding along (nnutils/mesh_net.py:514-516,
 * third_party/ext_utils/util_rot.py:27-37): pair_angle[i] = angle(Q[i], Q[i + M*K/2]) for i < M*K/2 (M even: first half of the batch =
 * frame t, second half = frame t') -- the values and gradients of lasr_geodesic_forward / _backward on (Q[:half], Q[half:]),
 * bit for bit, without their two launches; grad_quat = the fix-up's part + the distance's part. */
int lasr_bone_fixup_pair_forward(const float* quat, const float* trans, const float* depth, const float* rest_ts, float* rmat,
                                 float* tmat, float* pair_angle, int M, int H, int K, void* hip_stream);
int lasr_bone_fixup_pair_backward(const float* quat, const float* rest_ts, const float* grad_rmat, const float* grad_tmat,
                                  const float* grad_pair_angle, float* grad_quat, float* grad_trans, float* grad_depth,
                                  float* grad_rest, int M, int H, int K, void* hip_stream);

/*
 * Symmetric squared Chamfer distance of small point sets -- pytorch3d.loss.chamfer_distance()[0] as used on the bones' control
 * points at nnutils/mesh_net.py:500-503: a [N,P,3], b [N,Q,3] -> out[n] = mean_i min_j |a_i - b_j|^2 + mean_j min_i |b_j - a_i|^2
 * (the caller averages over n); nn_ab [N,P] / nn_ba [N,Q] int32 receive the nearest indices (first minimum) for the backward,
 * which writes grad_a [N,P,3] and grad_b [N,Q,3] for an upstream grad_out [N] by gathers (no atomics).
 */
int lasr_chamfer_forward(const float* a, const float* b, float* out, int* nn_ab, int* nn_ba, int N, int P, int Q,
                         void* hip_stream);
int lasr_chamfer_backward(const float* a, const float* b, const int* nn_ab, const int* nn_ba, const float* grad_out, float* grad_a,
                          float* grad_b, int N, int P, int Q, void* hip_stream);

/*
 * Mean shape of the batch, third_party/ext_nnutils/mesh_net.py:128-149 (symmetrize) + :171-185 (get_mean_shape): mean_v / tex
 * [H,Vp,3] (independent + right-half vertices per hypothesis) -> out_v / out_tex [R*H, Vp+S, 3]: the last S vertices are appended
 * once more mirrored (flip [3], e.g. (-1,1,1)), positions are multiplied by mask [Vp+S,3] (NULL = none; zeros pin the plane
 * vertices), colours go through a sigmoid, and the H meshes are tiled R times (one per image).  S = 0: no symmetry.
 * Backward: grad_mean_v / grad_tex_param [H,Vp,3] (either may be NULL), summed over the R copies in order.
 */
int lasr_mean_shape_forward(const float* mean_v, const float* tex, const float* flip, const float* mask, float* out_v,
                            float* out_tex, int R, int H, int Vp, int S, void* hip_stream);
int lasr_mean_shape_backward(const float* tex, const float* flip, const float* mask, const float* grad_v, const float* grad_tex,
                             float* grad_mean_v, float* grad_tex_param, int R, int H, int Vp, int S, void* hip_stream);

/*
 * Observed images of the texture losses, nnutils/mesh_net.py:364-366: fg = masks > 0; out[:n] = imgs * fg (object on black),
 * out[n:] = 1 - fg + imgs * fg (object on white).  imgs [n,3,P], masks [n,P] -> out [2n,3,P]; data only, no gradient.
 */
int lasr_obs_pair(const float* imgs, const float* masks, float* out, int n, int P, void* hip_stream);

/*
 * A training batch as one row gather: the input side of LASRTrainer.set_input, nnutils/train_utils.py:125-181, for a sequence
 * that is resident in HBM.  table [pairs, W] holds, per distinct frame pair, the model's batch
 * dictionary of that pair key after key (key k at columns [seg_off[k], seg_off[k] + seg_len[k]), frame t then frame t');
 * ids [B] (int64, device) selects the pairs; out receives key k's [B, seg_len[k]] block at out_off[k] -- pair-major, the
 * interleaved layout of train_utils.py:179-180.  One launch; n_keys <= LASR_GATHER_MAX_KEYS.  The offset arrays are HOST memory.
 */
/*
 * Everything LASR.forward does with the render between the rasteriser and the loss sum, in one pass over the [N,10,P] output of
 * the nine-attribute render (planes 0-2 texture colours, 3-5 own camera-space position, 6-8 the other frame's position, 9 alpha;
 * N = I*H images, image ij = i*H + j, first half of the batch = frame t, second half = frame t') and one pass back:
 *   flow reprojection + background mask (nnutils/mesh_net.py:87-104), silhouette table (:374-390), flow table + weighted error
 *   map + selection mask (:393-416), texture L1 table (:419-441), and the perceptual network's input pair
 *   rndpair [2N,3,P] = (render * alpha | render) (:436-441; pass NULL to skip it).
 * masks / occ [I,P]; flow_obs [I,C>=2,P] with image stride flow_obs_image_stride; img_obs / img_white [I,3,P]; pp [N,2], fl [N]:
 * image n projects its own position with (pp[n], fl[n]) (no gradient) and the other frame's with (pp[o], fl[o]), o = (n + N/2) % N.
 * Tables [I,H] are bit-identical to lasr_mask_loss_* / lasr_flow_loss_* / lasr_tex_loss_* on contiguous copies of the planes.
 * Backward: grad_px [N,10,P] is written whole (zeros on planes 3-5), grad_pp [N,2], grad_fl [N]; grad_rndpair may be NULL.
 * scratch: lasr_render_tables_scratch_floats(I,H,P) floats, carried from forward to backward.
 */
size_t lasr_render_tables_scratch_floats(int I, int H, int P);
int lasr_render_tables_forward(const float* px, const float* masks, const float* occ, const float* flow_obs,
                               long long flow_obs_image_stride, const float* img_obs, const float* img_white, const float* pp,
                               const float* fl, float l1tex_wt, float* mask_tab, float* flow_tab, float* tex_tab, float* flow_rd,
                               unsigned char* bgmask, float* flow_map, unsigned char* vis_mask, float* rndpair, float* scratch,
                               int I, int H, int P, void* hip_stream);
/* The same pass from the observed images themselves (round 6): imgs [I,3,P]; the object on black and on white (nnutils/mesh_net.py:364-366,
 * lasr_obs_pair's values) is formed inside and written to obs_pair_out [2I,3,P] (black first) -- what the backward and the perceptual
 * term take as img_obs / img_white.  One launch (lasr_obs_pair) and three input planes less than the call above; same tables. */
int lasr_render_tables_forward_imgs(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                    long long flow_obs_image_stride, const float* imgs, float* obs_pair_out, const float* pp,
                                    const float* fl, float l1tex_wt, float* mask_tab, float* flow_tab, float* tex_tab, float* flow_rd,
                                    unsigned char* bgmask, float* flow_map, unsigned char* vis_mask, float* rndpair, float* scratch,
                                    int I, int H, int P, void* hip_stream);
int lasr_render_tables_backward(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                long long flow_obs_image_stride, const float* img_obs, const float* img_white, const float* pp,
                                const float* fl, float l1tex_wt, const float* grad_mask_tab, const float* grad_flow_tab,
                                const float* grad_tex_tab, const float* grad_rndpair, const float* scratch, float* grad_px,
                                float* grad_pp, float* grad_fl, int I, int H, int P, void* hip_stream);

/*
 * What LASR.forward builds from the camera-space vertices before it calls the rasteriser (nnutils/mesh_net.py:298-311, :350-356;
 * geom_utils.py:27-34), in one launch + a one-block finish: verts_pre [N,V,3] = (pinhole(verts_cam; pp, fl) + eye) * (1,-1,1),
 * attrs [N,V,9] = (tex | verts_cam | verts_cam of image (n + N/2) % N), near_far [2] = (zmin - r/2, zmax + r/2), r = zmax - zmin
 * over all N meshes.  verts_cam / tex [N,V,3], pp [N,2], fl [N], eye: 3 HOST floats; scratch: 2 N floats.  N even.
 * Backward: grad_verts_cam (projection + both attribute uses), grad_tex, grad_pp [N,2], grad_fl [N], all overwritten.
 */
int lasr_raster_inputs_forward(const float* verts_cam, const float* tex, const float* pp, const float* fl, const float* eye,
                               float* verts_pre, float* attrs, float* near_far, float* scratch, int N, int V, void* hip_stream);
int lasr_raster_inputs_backward(const float* verts_cam, const float* fl, const float* grad_verts_pre, const float* grad_attrs,
                                float* grad_verts_cam, float* grad_tex, float* grad_pp, float* grad_fl, int N, int V,
                                void* hip_stream);

/*
 * The same stage per FACE CORNER, one launch each way (what the rasteriser consumes; replaces, for LASR.forward's own render,
 * lasr_raster_inputs_forward + the camera stage's `vertices - eye` (third_party/softras/soft_renderer/functional/look_at.py:6-62
 * with a constant eye on the -z axis, rotation = identity) + two lasr_face_gather_forward calls, and their three backward
 * launches):
 *   face_vertices[n,f,c,:] = ((pinhole(verts_cam[n, faces[f,c]]) + eye) * (1,-1,1)) - eye          [N,F,3,3]
 *   face_attrs[n,f,c,:]    = (tex | verts_cam | verts_cam of mesh (n + N/2) % N) at that vertex      [N,F,3,9]
 *   near_far               = as lasr_raster_inputs_forward
 * faces: int64 [N,F,3], or [1,F,3] with faces_shared = 1 (all meshes share the connectivity).  Backward: the vertex-centric sums
 * run over a CSR incidence structure the caller builds once per connectivity: inc_ptr int32 [N or 1, V+1], inc int32 [N or 1, 3F] =
 * corner ids (3 f + c) grouped by vertex, ASCENDING inside a vertex (the summation order of lasr_face_gather_backward).
 * scratch: lasr_raster_faces_scratch_floats(N, V, F) floats (either direction; block partials of the depth range / of the
 * intrinsics' gradients, folded by a second small launch).
 */
size_t lasr_raster_faces_scratch_floats(int N, int V, int F);
int lasr_raster_faces_forward(const float* verts_cam, const float* tex, const float* pp, const float* fl, const float* eye,
                              const long long* faces, int faces_shared, float* face_vertices, float* face_attrs, float* near_far,
                              float* scratch, int N, int V, int F, void* hip_stream);
int lasr_raster_faces_backward(const float* verts_cam, const float* fl, const int* inc_ptr, const int* inc, int faces_shared,
                               const float* grad_face_vertices, const float* grad_face_attrs, float* grad_verts_cam,
                               float* grad_tex, float* grad_pp, float* grad_fl, float* scratch, int N, int V, int F,
                               void* hip_stream);

#define LASR_GATHER_MAX_KEYS 24
int lasr_gather_rows(const float* table, long long W, int pairs, const long long* ids, int B, int n_keys,
                     const long long* seg_off, const long long* seg_len, const long long* out_off, float* out,
                     void* hip_stream);

/*
 * The optimisation-step tail, nnutils/train_utils.py:282-296: clip the mean-shape gradient to
 * norm max_norm_shape (1) and the encoder + code-predictor gradients jointly to max_norm_cam (10) with
 * torch.nn.utils.clip_grad_norm_'s coefficient min(1, max / (norm + 1e-6)); if ANY gradient element is NaN / Inf every gradient
 * becomes zero (the reference's zero_grad()) and the update still runs; then AdamW (decoupled weight decay, torch.optim.AdamW
 * arithmetic: p -= lr wd p; m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)) on
 * every tensor, and each tensor's step counter += 1.  Three launches over all tensors, no host synchronisation.
 *   table   device array, one row of 8 x 64 bit per tensor: {param, grad, exp_avg, exp_avg_sq, step (float*, may be 0), numel,
 *           group index, clip class (0 none, 1 mean shape, 2 camera networks)}
 *   chunks  device int32 [n_chunks, 2] = (table row, element offset), one per lasr_tail_chunk_elems() elements of a tensor
 *   partials device DOUBLE [n_chunks] scratch (sums of squares cannot overflow into a false NaN verdict); ctl device float [8]:
 *   ctl[6] counts the steps whose gradients were zeroed (the caller zeroes it once); out {clip coefficient shape, cam, all-finite flag,
 *           shape gradient norm after clipping, camera-network gradient norm before clipping, norm of all gradients}
 *   lr .. weight_decay: HOST arrays per parameter group (n_groups <= LASR_TAIL_MAX_GROUPS); bias_correction1/2 = 1 - beta^t of
 *           the step being taken, evaluated by the caller in double.
 */
#define LASR_TAIL_MAX_GROUPS 16
int lasr_tail_chunk_elems(void);
int lasr_tail_step(const void* table, const int* chunks, int n_chunks, double* partials, float* ctl, float max_norm_shape,
                   float max_norm_cam, const float* lr, const float* beta1, const float* beta2, const float* eps,
                   const float* weight_decay, const double* bias_correction1, const double* bias_correction2, int n_groups,
                   void* hip_stream);

/*
 * Plumbing around the raster calls.
 * lasr_fill_planes: dst [N, n_values, plane_elems], plane c of every image := values[c] (host array) -- the background fill the
 *   reference does before its forward kernel (third_party/softras/soft_renderer/functional/soft_rasterize.py:50-53: colour planes =
 *   background, alpha plane = 1).
 */
#define LASR_FILL_MAX_PLANES 16
int lasr_fill_planes(float* dst, const float* values, int n_values, int N, long long plane_elems, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LASR_OPS_H_ */
