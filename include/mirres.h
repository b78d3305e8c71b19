/* mirres.h — C ABI of libmirres.so: the MI355X (gfx950) engine behind the reference's ReSTIR path-tracing
 * operator surface (brabbitdousha/MIRReS-ReSTIR_Nerf_mesh, nerf/renderer_restir.py + nerf/ScreenSpaceReSTIR/...).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name starts with h_; all tensors are contiguous
 *     row-major fp32 / int32 exactly as the reference's torch tensors;
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no host sync) unless stated;
 *   - return value: 0 = ok, negative = MIRRES_E_*; nothing is allocated inside a call except in *_create and on the first
 *     use (or growth) of a pool kept in the context / BVH object: mirres_render's batch pool, the closest-hit redo lists;
 *   - pixelIndex = y * fx + x;  reservoir = (light_data f32[N,3], light_pdf f32[N], M i32[N], weight f32[N]).
 * Each entry point cites the reference interface it replaces (file:line relative to the reference repo).
 */
#ifndef MIRRES_H
#define MIRRES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIRRES_OK 0
#define MIRRES_E_ARG (-1)
#define MIRRES_E_HIP (-2)
#define MIRRES_E_STATE (-3)

typedef struct mirres_bvh mirres_bvh_t; /* owns the traversal layout + build workspace  */
typedef struct mirres_ctx mirres_ctx_t; /* owns per-frame-size ray queues + tables        */

/* ReSTIR constants: load_m_for_restir defines (renderer_restir.py:151-181) + in-shader #defines
 * (FinalShading.slang:7-9). max_bounce is a runtime parameter here (BASELINE config 5).                    */
typedef struct mirres_config {
    int light_tile_count;      /* 128  */
    int light_tile_size;       /* 1024 */
    int screen_tile_size;      /* 8    */
    int initial_light_samples; /* 32   */
    int initial_brdf_samples;  /* 1    */
    int max_history;           /* 20   */
    int neighbor_offset_count; /* 8192 */
    int neighbor_count;        /* 5    */
    float gather_radius;       /* 30   */
    int max_bounce;            /* 2    */
    float vis_near;            /* 0.01 */
} mirres_config_t;
void mirres_default_config(mirres_config_t* cfg);

const char* mirres_version(void);
const char* mirres_last_error(void);

/* ------------------------------------------------------------------ LBVH  (restirbvhWorker, renderer_restir.py:13-94) */
int mirres_bvh_create(mirres_bvh_t** out, int max_tris);
void mirres_bvh_destroy(mirres_bvh_t* bvh);
/* update_mesh + update_bvh (renderer_restir.py:25-94; kernels under nerf/bvhworkers). Writes the reference
 * node arrays LBVHNode_info i32[2T-1,3] / LBVHNode_aabb f32[2T-1,6] and the internal traversal layout.
 * sorted_codes i32[T,2] (code, elementIdx) may be NULL. No host synchronisation.                            */
int mirres_bvh_build(mirres_bvh_t* bvh, const float* vert, int V, const int32_t* tri, int T, int32_t* info, float* aabb,
                     int32_t* sorted_codes, void* stream);
/* Round 6. The private steering hierarchy of the shadow-ray / ordered closest-hit kernels (no reference counterpart: the reference traverses its LBVH as built; any
 * hierarchy over the same leaves gives the same answers) costs 0.2 ms (extended-Morton tree) + 0.9-1.4 ms (binned-SAH top) per build and pays for itself only on a long
 * frame: a caller that rebuilds every frame (render_stage1 does, nerf/renderer.py:975) builds with private_level = 1 and asks for the SAH top when the frame is long
 * enough (renderer_restir.py does: from ~1e8 pixel-samples). private_level: 0 = collapsed reference LBVH, 1 = extended-Morton tree, 2 = + SAH top, -1 =
 * MIRRES_PRIVATE_TREE / default 2 (what mirres_bvh_build does). mirres_bvh_upgrade completes a level-1 build to level 2 (pass the arrays of the build call; a no-op
 * on any other level); mirres_bvh_private_level reports the level of the last build.                                                                  */
int mirres_bvh_build_level(mirres_bvh_t* bvh, const float* vert, int V, const int32_t* tri, int T, int32_t* info, float* aabb,
                           int32_t* sorted_codes, int private_level, void* stream);
int mirres_bvh_upgrade(mirres_bvh_t* bvh, const float* vert, const int32_t* tri, const int32_t* info, const float* aabb, void* stream);
int mirres_bvh_private_level(mirres_bvh_t* bvh);
/* bvh_hit / bvh_hit_with_normal (utils/helperDi.slang:197-274, 313-395) over a batch of rays.
 * rays f32[n,8] = (ox,oy,oz,t_min, dx,dy,dz,t_max). mode 0: any-hit (early exit; only `hit` is written),
 * mode 1: closest by exhaustion in the reference's traversal order (hit,t,pos,normal,prim written; NULL skips).
 * mode 2: closest, same outputs bit for bit, via the front-to-back 4-wide fast path + reference-order recomputation of the rays whose
 *         result could depend on the visiting order (t <= 0 hits, exact ties); may grow an internal n-entry buffer on first use.
 * mode 3: occlusion as a conventional ray tracer answers it — some triangle is hit IN FRONT of the origin (t > 0); only `hit` is written.
 *         This is what nerf/render_dump.py:batch_intersector (:8-27) asks of the external `intersector` (intersects_closest(...)[0]).
 * mode 4: closest hit as a conventional ray tracer answers it — the nearest triangle met at t_min < t <= t_max (outputs as mode 1; no counters). The
 *         reference has no such query (its bvh_hit accepts triangles behind the origin); mirres_rasterize casts its near-plane rays with it.
 * counters u32[n,4] (popped, entered-internal, leaves-tested, stack-overflow) may be NULL.                   */
int mirres_bvh_trace(mirres_bvh_t* bvh, const float* rays, int n, int mode, int32_t* hit, float* t, float* pos, float* normal,
                     int32_t* prim, uint32_t* counters, void* stream);

/* ------------------------------------------------------------------ context (load_m_for_restir, renderer_restir.py:148-228) */
int mirres_ctx_create(mirres_ctx_t** out, int fx, int fy, const mirres_config_t* cfg);
void mirres_ctx_destroy(mirres_ctx_t* ctx);
/* createNeighborOffsetTexture (make_sampleable.slang:186-205) / 127 -> f32[count,2]; ctx keeps a copy.      */
int mirres_neighbor_offsets(mirres_ctx_t* ctx, float* out, void* stream);
/* totals since the last reset (host; synchronises): u64[16] = rays_any, rays_closest, then (popped, entered, leaves) of the any-hit
 * kernel and (popped, entered, leaves) of the closest-hit kernel — the node counts only advance while instrument bit 0 is set —
 * [8] / [9] deepest private traversal stack seen by the shadow-ray / ordered closest-hit kernel (instrument bit 0), [10] rays the
 * ordered closest-hit kernel handed to the reference-order kernel (instrument bit 0), [11] private-stack overflows of the
 * shadow-ray kernel (always counted; provably 0, bvh_trace.hip MR_ANY_STACK), [12] shadow rays of the spatial pass that were
 * not traced because the merge cannot see their answer (light reservoir with luminance 0; instrument bit 0; they are
 * included in rays_any), [13..15] shadow-ray kernel, instrument bit 0: wave iterations, wave iterations that ran the leaf
 * branch, leaf records fetched (how often a wave pays for the leaf branch and for how many lanes: bench.py roofline.leaf_branch).
 * A private-stack overflow also sets a sticky flag that makes mirres_render / mirres_bvh_trace return MIRRES_E_STATE afterwards.   */
int mirres_ctx_stats(mirres_ctx_t* ctx, uint64_t* h_out, int reset);
/* instrument bit 0: traversal kernels count visited nodes into the stats (slower kernels); bit 1: every traversal launch is
 * bracketed by HIP events on its own stream so that mirres_ctx_trace_time can report per-kernel durations; bit 2 (with bit 0):
 * shadow rays are counted by the reference-order traversal (bvh_hit's own visits, helperDi.slang:197-274) instead of the
 * production kernel's. Instrumented frames run on one stream.                                                           */
int mirres_ctx_set_instrument(mirres_ctx_t* ctx, int on);
/* sums the event-timed traversal launches since the last call (host; synchronises): ms and launch counts for the any-hit
 * and the closest-hit kernel.                                                                                         */
int mirres_ctx_trace_time(mirres_ctx_t* ctx, double* h_ms_any, int* h_n_any, double* h_ms_closest, int* h_n_closest);

/* ------------------------------------------------------------------ environment light */
/* make_sampleable (GenerateLightTiles.py:4-29 + make_sampleable.slang:34-86). env_tex f32[Hc*Wc,3] is the
 * vertically flipped, flattened map (renderer_restir.py:305-311). Outputs pdf[Hc*Wc], cdf[Hc*(Wc+1)], mpdf[Hc], mcdf[Hc+1]. */
int mirres_env_make_sampleable(const float* env_tex, int Wc, int Hc, float* pdf, float* cdf, float* mpdf, float* mcdf, void* stream);
/* GenerateLightTiles (GenerateLightTiles.py:31-52, GenerateLightTiles.slang:16-62).                          */
int mirres_light_tiles(mirres_ctx_t* ctx, const float* env_tex, int Wc, int Hc, const float* pdf, const float* cdf, const float* mpdf,
                       const float* mcdf, uint32_t frameIndex, float* light_data, int32_t* light_uv, float* light_inv_pdf, void* stream);

typedef struct mirres_env { const float* tex; int Wc, Hc; const float *pdf, *cdf, *mpdf, *mcdf; } mirres_env_t;
typedef struct mirres_gbuf { const float *occ, *pos, *normal_depth, *brdf, *ray_dir; } mirres_gbuf_t; /* [N],[N,3],[N,4],[N,3],[N,3] */
typedef struct mirres_res { float* light_data; float* light_pdf; int32_t* M; float* weight; } mirres_res_t;

/* ------------------------------------------------------------------ reservoir passes */
/* restirbvhWorker.InitialResampling_ (renderer_restir.py:96-114; InitialResampling.slang:151-295)            */
int mirres_restir_initial(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res,
                          const float* light_data, const float* light_inv_pdf, uint32_t frameIndex, void* stream);
/* TemporalResampling (Resampling.py:26-45; TemporalResampling.slang:23-135). motion may be NULL (= zeros).   */
int mirres_restir_temporal(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_gbuf_t* prev_g,
                           const mirres_res_t* res, const mirres_res_t* prev_res, const float* motion, uint32_t frameIndex, void* stream);
/* restirbvhWorker.SpatialResampling_ (renderer_restir.py:116-131; SpatialResampling.slang:178-322)           */
int mirres_restir_spatial(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_gbuf_t* g, const mirres_res_t* res,
                          const mirres_res_t* prev_res, const float* neighbor_offsets, uint32_t frameIndex, void* stream);
/* restirbvhWorker.EvaluateFinalSamples_get_vis (renderer_restir.py:133-146; EvaluateFinalSamples.slang:84-124) */
int mirres_restir_final_vis(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const float* pos, const mirres_res_t* res, float* vis_map, void* stream);
/* EvaluateFinalSamples_di.forward/backward (Resampling.py:94-143; EvaluateFinalSamples.slang:129-188)        */
int mirres_restir_eval_final(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_res_t* res, const float* vis_map, float* final_dir,
                             float* final_dist, float* final_Li, void* stream);
int mirres_restir_eval_final_bwd(mirres_ctx_t* ctx, const mirres_env_t* env, const mirres_res_t* res, const float* vis_map,
                                 const float* grad_final_Li, float* grad_env /*[Hc*Wc,3], accumulated*/, void* stream);

/* ------------------------------------------------------------------ shading / path tracing */
/* FinalShading.forward/backward (Resampling.py:145-214; FinalShading.slang:14-109)                           */
int mirres_final_shading(mirres_ctx_t* ctx, const mirres_env_t* env, const float* occ, const float* normal, const float* ray_dir,
                         const float* kd, const float* rough_metal, const float* final_dir, const float* final_dist, const float* final_Li,
                         float* color, float* diff_light, float* spec_light, void* stream);
int mirres_final_shading_bwd(mirres_ctx_t* ctx, const float* occ, const float* normal, const float* ray_dir, const float* kd,
                             const float* rough_metal, const float* final_dir, const float* final_dist, const float* final_Li,
                             const float* g_color, const float* g_diff, const float* g_spec, float* g_normal, float* g_kd,
                             float* g_rough_metal, float* g_final_Li, void* stream);
typedef struct mirres_path {
    const float *occ, *pos, *normal, *ray_dir, *kd, *rough_metal; /* vertex inputs                              */
    float* prd;                                                   /* f32[N,5] throughput rgb, specularBounce, stop */
    float *new_pos, *new_ray_d, *new_occ, *new_normal;            /* next-vertex outputs                        */
} mirres_path_t;
/* process_new_dir_for_pt (Resampling.py:216-232; FinalShading.slang:113-265)                                  */
int mirres_pt_new_dir(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_path_t* p, uint32_t frameIndex, uint32_t bounce_count, void* stream);
/* indirect_one_hit_divided_no_grad (Resampling.py:254-273; FinalShading.slang:641-1009). When acc_* are non-NULL
 * the results are ADDED to them instead of overwriting color/diff/spec (fuses renderer_restir.py:420-422).     */
int mirres_pt_bounce(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_env_t* env, const mirres_path_t* p, uint32_t frameIndex,
                     uint32_t bounce_count, float* color, float* diff_color, float* spec_color, void* stream);

/* ------------------------------------------------------------------ denoiser */
/* EAWDenoise_run.forward / EAWDenoise_run_no_di (Denoising.py:10-60; EAWDenoise.slang:50-302)                 */
int mirres_eaw(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color,
               const float* normal, const float* pos, float* out, void* stream);
/* process_normal_ao (EAWDenoise.slang:591-651, launched at nerf/renderer.py:1153-1158 when --lambda_extra_kd > 0): per foreground pixel the weight
 * clamp(50 (1 - mean over the 8 x 8 window's foreground pixels of clamp(n_q . n_p, 0, 1)), 0, 1) written to the three channels of out_ao f32[N,3];
 * background pixels get 0. ray_dir is an argument of the reference kernel that it never reads and is not taken here.                             */
int mirres_normal_ao(int fx, int fy, const float* occ, const float* normal, float* out_ao, void* stream);
/* EAWDenoise_run.backward (Denoising.py:30-48): grads w.r.t. colour, normal and position are ACCUMULATED.      */
int mirres_eaw_bwd(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color,
                   const float* normal, const float* pos, const float* grad_out, float* g_color, float* g_normal, float* g_pos, void* stream);
/* The same adjoint computed as a gather (two passes, no atomics, deterministic; scratch4 f32[N,4] is overwritten). The three gradient buffers are
 * accumulated into like mirres_eaw_bwd's.                                                                                                      */
int mirres_eaw_bwd_gather(int fx, int fy, int step_width, float c_phi, float n_phi, float p_phi, const float* occ, const float* color, const float* normal,
                          const float* pos, const float* grad_out, float* scratch4, float* g_color, float* g_normal, float* g_pos, void* stream);

/* bilateral_denoiser (nerf/renderutils/ops.py:173-211; c_src/denoising.cu:14-130): the alternative denoiser of run_restir_di_with_pt
 * (--use_bi_de, renderer_restir.py:529-541). sigma = max(2 * factor, 1e-4); window radius 2 ceil(2.5 sigma) + 1. col f32[N,3],
 * nrm f32[N,3] (normalised inside: safe_normalize, ops.py:168), zdz f32[N,2] = (z, |dz|); out f32[N,4] = (sum w col, max(sum w, 1e-4)) —
 * the caller divides (ops.py:198). scratch f32[N,8] is overwritten.                                                              */
int mirres_bilateral(int fx, int fy, float sigma, const float* col, const float* nrm, const float* zdz, float* scratch, float* out4, void* stream);
/* _bilateral_denoiser_func.backward (ops.py:181-185): col_grad f32[N,3] from grad_out f32[N,4] (its 4th channel carries no gradient to col). */
int mirres_bilateral_bwd(int fx, int fy, float sigma, const float* nrm, const float* zdz, const float* grad_out4, float* scratch, float* col_grad,
                         void* stream);

/* dump_render (nerf/render_dump.py:84-133) = dump_render_run_mesh (:136-215) + GGX_specular (:32-65) + get_light_rgbs (:70-82) +
 * batch_intersector (:8-27): the reference's direct-lighting renderer without ReSTIR (`--use_brdf` alone, nerf/renderer.py:1131-1149;
 * BASELINE configs[0]). n surface points (pos, normal, albedo, roughness, fresnel, rays_d: f32[n,3] each; roughness / fresnel are the
 * 3-channel repeats of renderer.py:1134-1135), env_map f32[env_h,env_w,3], the fixed light set of generate_envir_map_dir
 * (nerf/render_helper.py:8-26): light_dirs f32[L,3], light_area_weight f32[L] (may be NULL when equal_areas != 0 — sample_method
 * 'stratifed_sample_equal_areas'). Occlusion = mode 3 of mirres_bvh_trace from pos + 0.001 dir for every light with clamped cosine > 1e-6.
 * light_rgbs f32[L,3] receives get_light_rgbs; out_rgb (clamped to [0,1] when clamp_rgb, as dump_render does), out_diff, out_spec f32[n,3].
 * Works through the points in chunks of <= 2^24 (point, light) pairs; the chunk's ray pool (40 B per pair) is kept in the BVH object.  */
int mirres_dump_render(mirres_bvh_t* bvh, int n, int L, const float* pos, const float* normal, const float* albedo, const float* roughness,
                       const float* fresnel, const float* rays_d, const float* env_map, int env_h, int env_w, const float* light_dirs,
                       const float* light_area_weight, int equal_areas, int clamp_rgb, float* light_rgbs, float* out_rgb, float* out_diff,
                       float* out_spec, void* stream);

/* The step in front of the path: what render_stage1 gets from nvdiffrast (an un-vendored dependency of the reference).
 * mirres_raster_raycast stands in for dr.rasterize (nerf/renderer.py:983): primary rays f32[n,8] (as mirres_bvh_trace) are cast through the BVH
 * (closest hit) and rast f32[n,4] receives nvdiffrast's raster record (u, v, t, triangle_id + 1), zeros where nothing is hit; vert f32[V,3] /
 * tri i32[T,3] are the arrays the BVH was built from. mirres_interpolate(_bwd) = dr.interpolate without attribute derivatives (:985, :996,
 * :998): out f32[n,C] = u a[i0] + v a[i1] + (1-u-v) a[i2] (0 where triangle_id = 0); the backward ACCUMULATES into g_attr f32[V,C] and writes
 * g_uv f32[n,2] (either may be NULL). dr.texture and dr.antialias are not provided.                                                      */
int mirres_raster_raycast(mirres_bvh_t* bvh, const float* rays, int n, const float* vert, const int32_t* tri, float* rast, void* stream);
/* dr.rasterize(glctx, pos_clip, tri, (H, W)) itself (nerf/renderer.py:983; with rast_db as :1074 hands it to dr.interpolate): h_mvp = HOST float[16],
 * row-major, clip = mvp * (world, 1) — the matrix behind pos_clip = pad(vertices) @ mvp^T (:981), a perspective projection; h_eye = HOST float[3], the
 * world-space point with x_c = y_c = w_c = 0 (the eye: every pixel's line passes through it).  Pixel (ix, iy) of the W x H
 * image looks along NDC ((2 ix + 1) / W - 1, (2 iy + 1) / H - 1); its world-space line is cast through the BVH (the world-space vert / tri it was built
 * from).  rast f32[H*W,4] = (u, v, z/w, triangle_id + 1) with perspective-correct barycentrics (weights of v0, v1) and the clip-space depth of the hit;
 * rast_db f32[H*W,4] = (du/dX, du/dY, dv/dX, dv/dY) per pixel, or NULL.  The ray starts on the near plane (what lies in front of it is clipped away and hides nothing); hits beyond the far plane give an empty record.   */
int mirres_rasterize(mirres_bvh_t* bvh, const float* vert, const int32_t* tri, const float* h_mvp, const float* h_eye, int W, int H,
                     float* rast, float* rast_db, void* stream);
int mirres_interpolate(const float* attr, int C, const float* rast, const int32_t* tri, int n, float* out, void* stream);
int mirres_interpolate_bwd(const float* attr, int C, const float* rast, const int32_t* tri, int n, const float* g_out, float* g_attr, float* g_uv, void* stream);
/* dr.texture(tex, uv, filter_mode='linear', boundary_mode='clamp') (nerf/renderer.py:1004, 1008: the jittered taps of the smoothness
 * regularisers): tex f32[H,W,C], uv f32[n,2] in [0,1] (texel centres at (i + 0.5) / W) -> out f32[n,C]; the backward ACCUMULATES into g_tex. */
int mirres_texture2d(const float* tex, int H, int W, int C, const float* uv, int n, float* out, void* stream);
int mirres_texture2d_bwd(int H, int W, int C, const float* uv, int n, const float* g_out, float* g_tex, void* stream);
/* dr.antialias (nerf/renderer.py:1184-1206; nvdiffrast is un-vendored: the published algorithm, csrc/antialias.hip): color f32[H*W,C], rast f32[H*W,4]
 * (mirres_raster_raycast's record: .z orders the two triangles of a pixel pair by distance, .w = triangle id + 1), pos_clip f32[V,4] clip-space
 * vertex positions (pixel (i, j)'s centre is NDC ((2i + 1) / W - 1, (2j + 1) / H - 1)), tri i32[T,3], opp i32[T,3] the vertex across each edge
 * (v_k, v_k+1) in the neighbouring triangle or -1 (topology: depends on `tri` only).  out f32[H*W,C], written without atomics (deterministic).
 * _bwd: g_color f32[H*W,C] (overwritten) and / or g_pos f32[V,4] (ACCUMULATED with atomics; scaled by pos_gradient_boost); either may be NULL. */
int mirres_antialias(int W, int H, int C, const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* opp,
                     float* out, void* stream);
int mirres_antialias_bwd(int W, int H, int C, const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* opp,
                         const float* g_out, float* g_color, float* g_pos, float pos_gradient_boost, void* stream);

/* prepare_shading_normal (nerf/renderutils/ops.py:100-163; c_src/normal.cu): the shading normal render_stage1 hands to the path
 * (nerf/renderer.py:1013). All inputs f32[n,3] (broadcast inputs expanded by the caller); out f32[n,3]. The backward writes the six input
 * gradients (any may be NULL).                                                                                                   */
int mirres_prepare_shading_normal(long long n, const float* pos, const float* view_pos, const float* perturbed_nrm, const float* smooth_nrm,
                                  const float* smooth_tng, const float* geom_nrm, int two_sided_shading, int opengl, float* out, void* stream);
int mirres_prepare_shading_normal_bwd(long long n, const float* pos, const float* view_pos, const float* perturbed_nrm, const float* smooth_nrm,
                                      const float* smooth_tng, const float* geom_nrm, int two_sided_shading, int opengl, const float* dout,
                                      float* g_pos, float* g_view_pos, float* g_perturbed_nrm, float* g_smooth_nrm, float* g_smooth_tng,
                                      float* g_geom_nrm, void* stream);

/* ------------------------------------------------------------------ material field (MLPTexture3D, render_helper.py:53-124) */
typedef struct mirres_matnet {
    const uint16_t* grid_f16; /* fp16 hash-grid table, 6 299 960 x 2 entries (tcnn layout)                      */
    const float *w0, *w1, *w2; /* torch Linear weights [32,32],[32,32],[6,32], row-major [out,in]                */
    float aabb_min[3], aabb_max[3], out_min[6], out_max[6];
} mirres_matnet_t;
int mirres_matnet_grid_entries(void);                      /* 6 299 960                                        */
int mirres_matnet_pack_grid(const float* params_f32, uint16_t* grid_f16, int64_t n, void* stream);
/* MLPTexture3D.sample / sample_no_di: pos f32[n,3] -> out f32[n,6]; enc_out (fp16 bits [n,32]) may be NULL.     */
int mirres_matnet_fwd(const mirres_matnet_t* m, const float* pos, int n, float* out, uint16_t* enc_out, void* stream);
/* _MLP.forward on precomputed encodings (render_helper.py:43-44, 100-104): enc fp16 bits [n,32] -> out f32[n,6] (sigmoid + range applied).
 * The MFMA-tiled kernel (v_mfma_f32_32x32x2_f32, sixteen K = 2 steps per layer in ascending k: the fp32 fmaf chain of mirres_matnet_fwd, bit for bit). */
int mirres_matnet_mlp(const mirres_matnet_t* m, const uint16_t* enc, int n, float* out, void* stream);
/* renderer_restir.py:398-408 fused: evaluate where occ>=0.5 and scatter kd / (roughness, metallic) in place.    */
int mirres_matnet_scatter(const mirres_matnet_t* m, const float* occ, const float* pos, int n, float* kd, float* rough_metal,
                          int use_scale, const float* h_scale3, void* stream);
/* backward of sample(): grads to the fp32 master grid (atomic, accumulated), the MLP weights (accumulated) and, when g_pos != NULL,
 * the sample positions f32[n,3] (overwritten; tcnn's HashGrid input gradient x 1/(aabb_max-aabb_min), zero where the clamp to the
 * AABB is active - render_helper.py:93-98).                                                                       */
int mirres_matnet_bwd(const mirres_matnet_t* m, const float* pos, int n, const float* grad_out, float* g_params_f32, float* g_w0,
                      float* g_w1, float* g_w2, float* g_pos, void* stream);

/* ------------------------------------------------------------------ whole frame: run_restir_di_with_pt (renderer_restir.py:473-550) */
typedef struct mirres_render_args {
    int spp; uint32_t random_offset; /* np.random.randint(2**20) in the reference (renderer_restir.py:245)     */
    int use_scale; float scale[3];
    const float* env_map; int Wc, Hc; /* f32[Hc,Wc,3] as passed by the caller (not flipped)                     */
    float* occ;                       /* f32[N]   modified in place (renderer_restir.py:484-485)                */
    const float *normal, *depth, *kd, *rough_metal, *ray_dir, *pos;
    const mirres_matnet_t* mat;       /* NULL -> const_kd / const_rm at indirect hits                           */
    float const_kd[3], const_rm[2];
    int denoise_iter, step_width; float c_phi, n_phi, p_phi;
    float* outs[6];                   /* final_color, den_diffuse, den_spec, den_indirect, den_indirect_diff, den_indirect_spec */
    float* tape;                      /* f32[samples, N, 8] or NULL: per sample and pixel the final (spatially merged) reservoir and its visibility
                                         {light_data.xyz inv_pdf | M weight vis 0}, what mirres_render_bwd differentiates through        */
    const float* gb_depth;            /* f32[N,2] (z, |dz|) or NULL: non-NULL selects the bilateral denoiser with factor 2 (:529-541) instead of EAW */
    int spp_begin, spp_end;           /* multi-GPU spp sharding: render samples [spp_begin, spp_end) and skip the
                                         average/denoise/composite (raw sums are left in outs[0..5]); 0,0 = all  */
    /* Multi-GPU strip sharding (exact: bit-identical to one GPU for the rows a rank owns). The context is created for the rank's
     * LOCAL frame = its own rows plus up to 30 halo rows (the spatial gather radius) on either side; all per-pixel inputs cover the
     * local frame (which may extend below the image by rows the caller padded with background, occ = 0, so that strips of different
     * heights share a context size). strip_full_fy = height of the whole frame (0 = no strip sharding), strip_y_off = global row of local row 0,
     * [own_y0, own_y1) = the local rows this rank owns. Halo rows are only read (G-buffer, reservoirs); `halo` is called once per
     * sample, on the host while the frame is being enqueued, between temporal and spatial reuse: it must enqueue — on the same
     * stream — the exchange that fills the halo rows of `records` (packed reservoirs, f32[local pixels, 8]) with the neighbouring
     * ranks' own rows and sends this rank's border rows. Like a spp slice, a strip leaves raw sums in outs[0..5].                */
    int strip_full_fy, strip_y_off, own_y0, own_y1;
    int (*halo)(void* user, float* records, int sample, void* stream);
    void* halo_user;
    /* strip_overlap != 0 (round 4; needs `halo`): the spatial pass of every sample runs in two parts. `halo` is called with a SIDE stream (ordered after the
     * temporal pass) and must enqueue the exchange there; meanwhile the caller's stream does the spatial pass of the INTERIOR rows — the own rows at least
     * gather_radius away from a strip edge with a neighbouring rank, whose neighbours are all own rows — and only then waits for the exchange and does the border
     * rows. Same results bit for bit; the exchange leaves the per-sample critical chain at the price of three more launches per sample (and the next sample's
     * temporal merge is not fused into the resolve kernel). Strips too short to have interior rows fall back to the in-line exchange.                       */
    int strip_overlap;
    /* Native exchange (round 6): halo_comm = a communicator of mirres_comm_create, or NULL. When set, mirres_render issues the per-sample exchange itself — for each of
     * the halo_n (<= 2) neighbouring ranks halo_peer[k] one ncclSend of the local rows [halo_send0[k], halo_send1[k]) and one ncclRecv into the local rows
     * [halo_recv0[k], halo_recv1[k]) of the packed reservoirs, inside one ncclGroupStart / ncclGroupEnd on the chain's stream — and `halo` is not called: no Python on
     * the per-sample path (the callback costs 56-63 us of host time per sample in the median over RCCL, 240-330 us in the mean: profiles/r06_halo_host_cost*.txt).
     * halo_time_stride > 0: every halo_time_stride-th exchange is bracketed by events on the stream; mirres_ctx_halo_time sums them (the strip's own busy time =
     * its render time minus the time spent inside the exchanges, which is where a rank waits for its neighbours: dist.StripBalancer).                              */
    void* halo_comm;
    int halo_n, halo_peer[2], halo_send0[2], halo_send1[2], halo_recv0[2], halo_recv1[2];
    int halo_time_stride;
} mirres_render_args_t;
int mirres_render(mirres_ctx_t* ctx, mirres_bvh_t* bvh, const mirres_render_args_t* a, void* stream);
/* The library's own RCCL communicator for the native halo exchange (no reference counterpart: /root/reference is single-GPU). librccl is dlopen-ed —
 * `librccl_path` (may be NULL / empty) is tried first, the copy already loaded into the process wins. mirres_comm_unique_id: ncclGetUniqueId into 128 bytes (rank 0; the
 * caller distributes them, e.g. with torch.distributed.broadcast); mirres_comm_create: ncclCommInitRank (collective over the `world` ranks, current device);
 * mirres_ctx_halo_time: blocks until the last recorded exchange has finished, returns the summed milliseconds of the bracketed exchanges of the last mirres_render on
 * this context and their number, and forgets them.                                                                                                    */
int mirres_comm_unique_id(const char* librccl_path, void* id128);
int mirres_comm_create(void** comm, const char* librccl_path, const void* id128, int world, int rank);
void mirres_comm_destroy(void* comm);
int mirres_ctx_halo_time(mirres_ctx_t* ctx, double* ms, int* exchanges);
/* Backward of the frame's DIRECT lighting sums w.r.t. what the reference differentiates (EvaluateFinalSamples_di.backward +
 * FinalShading.backward summed over the samples, Resampling.py:116-214): from the cotangents of the three direct sums (total colour,
 * diffuse light, specular light — f32[N,3] each) and the tape of the forward call (`samples` samples), gradients w.r.t. normal [N,3],
 * kd [N,3], (roughness, metallic) [N,2] (overwritten) and the environment texels f32[Hc*Wc,3] in the caller's (unflipped) layout
 * (accumulated). The indirect sums carry no gradient (process_path_tracing_divided_no_grad). `a` = the forward call's arguments.       */
int mirres_render_bwd(mirres_ctx_t* ctx, const mirres_render_args_t* a, int samples, const float* g_color, const float* g_diff,
                      const float* g_spec, float* g_normal, float* g_kd, float* g_rough_metal, float* g_env, void* stream);
/* Sizes the context's batch pool ahead of the first frame (no reference counterpart: load_m_for_restir allocates the reference's persistent buffers at
 * start-up, renderer_restir.py:189-217; here the K-sample pool — ~670 B per sample slot, 55 GB for 32 samples of a 1600 x 1600 frame — would otherwise
 * be allocated inside the first mirres_render of a frame size, a device synchronisation and a multi-GB hipMalloc + clear in the middle of the first
 * timed frame).  samples_per_batch = 0: the default batch (MIRRES_PT_BATCH, 32).  Returns the batch size actually reserved (smaller when the device
 * cannot spare the request) or a negative error code.                                                                                        */
int mirres_ctx_reserve(mirres_ctx_t* ctx, int samples_per_batch);
/* second half of run_restir_di_with_pt (:507-549) on already-summed accumulators (after an all-reduce).         */
int mirres_render_finish(mirres_ctx_t* ctx, const mirres_render_args_t* a, float* sums[6], void* stream);

/* Self-check of the arithmetic the shading kernels are built on (no reference counterpart: the reference relies on nvcc's IEEE division).
 * The shading translation units divide and take square roots with short instruction sequences (csrc/device_math.hpp, mr_div / mr_rcp /
 * mr_sqrt) that must return the bits of the IEEE-754 operations. This runs them against the compiler's correctly rounded `a / b`, `1 / b`
 * and `sqrtf` on the device: every significand pair (a, b) in [1, 2) x S, S = 2^log2_b significands spread over [1, 2) plus the last 256
 * (log2_b = 23: all 2^46 pairs, ~50 s on an MI355X), every reciprocal of [1, 2), and every square root of [1, 4).
 * out[0] = pairs tested, out[1..3] = mismatches of mr_div, mr_rcp, mr_sqrt (all must be 0). Blocks until done.                          */
int mirres_selfcheck_arith(int log2_b, unsigned long long out[4], void* stream);

/* The transcendental functions of the path on the device (no reference counterpart: the reference calls CUDA's math library in
 * utils/lightDi.slang:119-132,181-209,312-330, utils/brdf.slang:76-124, EAWDenoise.slang:50-302, res.slang:53-61; here they are the fixed
 * arithmetic of include/mirres_fmath.h, the same header the CPU oracle includes).  fn: 0 sin, 1 cos, 2 acos, 3 exp, 4 exp2, 5 pow5,
 * 6 x^8, 7 x^128, 8 sigmoid (one argument, `a`); 16 atan2(a, b); 17 b / a with the short division of the shading kernels (mr_div);
 * 18 the short square root (mr_sqrt).
 * mirres_fmath_eval: out[i] = fn(a[i], b[i]) for n device floats (b may be NULL for one-argument functions).
 * mirres_fmath_checksum: over the argument bit patterns first .. first + count - 1 (for fn >= 16 the second argument is a fixed hash of the
 * first), *out = sum of result_bits * (2 * argument_bits + 1) mod 2^64 with NaN results canonical — order-free, so the host can form the same
 * sum from the same header (oracle/fmath_check.cpp) and tests/test_gpu_fmath.py can compare device and host over all 2^32 arguments. Blocks. */
int mirres_fmath_eval(int fn, const float* a, const float* b, float* out, long long n, void* stream);
int mirres_fmath_checksum(int fn, unsigned int first, unsigned long long count, unsigned long long* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIRRES_H */
