/*
 * pbr_hip.h -- C ABI of libpbr_hip.so: fused Cook-Torrance evaluation of PBR
 * material maps on AMD MI355X (gfx950), plus the map conversions the path pulls in.
 *
 * This is the drop-in boundary of the build.  The reference
 * (giuvecchio/PyPBR) has no FFI: its hot path sits behind a Python callable,
 *     pypbr.models.CookTorranceBRDF.forward      pypbr/models/cooktorrance.py:68-182
 * and the entry points below are what a native binding for that path binds
 * (INTEGRATION.md shows the ctypes stub a PyPBR maintainer would add).  Plain
 * pointers and sizes only -- no torch types.  All map pointers are DEVICE pointers;
 * light/view parameters are HOST values (they travel in the kernel-argument
 * segment, i.e. in SGPRs: the wave-uniform broadcast is free).
 *
 * Threading: every call only enqueues work on `stream` (a hipStream_t passed as
 * void*, NULL = the null stream); no call synchronises with the host or allocates
 * device memory, and a call's behaviour is a function of its arguments (same
 * contract as the reference: synchronous-looking, stateless, re-entrant) -- with ONE
 * documented exception: pbr_set_tuning below is a process-global test / bench /
 * profiling hook (atomic words) that changes which schedule later calls of the whole
 * process pick.  Results never depend on it, only speed; product code leaves it alone
 * and passes per-call settings through pbr_render_desc.tuning instead.  Inputs are
 * never written.
 *
 * Map layout (pypbr/materials/base.py: maps are (C,H,W) float32, channel-first):
 * planar [B][C][H][W], rows contiguous (row stride == width).  `batch_stride`
 * and `channel_stride` are in ELEMENTS, so a row band of a taller map, or one
 * map shared by the whole batch (batch_stride = 0), is passed without a copy.
 */
#ifndef PBR_HIP_H
#define PBR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBR_HIP_ABI_VERSION 8      /* 8: pbr_cook_torrance_blend_backward over tiled maps (map-sized gradients), pbr_blend_backward_serves; 7: folded gradients of tiled maps (pbr_cook_torrance_backward_folded), the loss step over tiled maps, 12 schedule knobs (was 23); 6: pbr_render_desc.tuning (per-call schedule knobs; pbr_set_tuning demoted to a process-global test hook); 5: pbr_render_desc.device_params (view / light / intensity read from device memory); 4: light_size follows Python truthiness, gradients of the map ops */
#define PBR_MAX_LIGHTS 16

/* ---- status codes (negative = caller error, positive = HIP runtime error code + 1000) */
enum {
    PBR_OK = 0,
    PBR_ERR_NULL_MAP = -1,        /* a required map pointer is NULL */
    PBR_ERR_WORKFLOW = -2,        /* neither metallic nor specular: cooktorrance.py:115-118 ValueError */
    PBR_ERR_LIGHT_TYPE = -3,      /* cooktorrance.py:62-65 ValueError */
    PBR_ERR_SHAPE = -4,           /* non-positive extent, band outside the map, > PBR_MAX_LIGHTS */
    PBR_ERR_DTYPE = -5,
    PBR_ERR_CHANNELS = -6,        /* base.py:219 "Normal map must have 2 or 3 channels." */
    PBR_ERR_NO_DEVICE = -7,
    PBR_ERR_UNSUPPORTED = -8      /* valid request this build does not implement */
};

enum { PBR_F32 = 0, PBR_F16 = 1 };                 /* storage type of maps */
enum { PBR_LIGHT_DIRECTIONAL = 0, PBR_LIGHT_POINT = 1 };   /* cooktorrance.py:61-65 */
enum {
    PBR_WORKFLOW_METALLIC = 0,    /* BasecolorMetallicMaterial: cooktorrance.py:103-107 */
    PBR_WORKFLOW_SPECULAR = 1,    /* DiffuseSpecularMaterial:   cooktorrance.py:108-114 */
    PBR_WORKFLOW_CONVERTED = 2    /* metallic maps, converted in-kernel by
                                     to_diffuse_specular_material (metallic.py:71-120),
                                     rendered in the specular workflow (config 3) */
};

typedef struct pbr_map {
    const void *data;             /* device pointer, NULL = map absent */
    int64_t batch_stride;         /* elements between materials */
    int64_t channel_stride;       /* elements between channel planes */
} pbr_map;

/* ---- schedule knobs ------------------------------------------------------------------
 * HOW a launch is scheduled (never WHAT it computes: every setting gives bit-identical results).  Each knob has a built-in rule
 * (measured on MI355X); a caller that knows better passes a pbr_tuning with the descriptor -- per call, caller-owned,
 * nothing outlives the call.  Entries left at PBR_TUNE_UNSET follow the rule (or the process-wide test hook pbr_set_tuning). */
enum {
    PBR_TUNE_RESERVED_0 = 0,            /* (ABI <= 7: PBR_TUNE_NONTEMPORAL.  The streaming hints are rules since ABI 8 -- on for vector lanes, stores only in the repeat-inner kernels --
                                           and the instantiations only this knob reached are not built; the slot is ignored) */
    PBR_TUNE_BLOCK_LOG2 = 1,            /* workgroup = 1 << value lanes (6..8); 0 = rule */
    PBR_TUNE_F16_VEC = 2,               /* pixels per lane for fp16 maps (4 | 8) */
    PBR_TUNE_LDS_BYTES = 3,             /* unused dynamic LDS per workgroup of the render kernel: an occupancy governor (-1 = rule) */
    PBR_TUNE_RESERVED_4 = 4,            /* (ABI <= 7: PBR_TUNE_BWD_VEC.  Pixels per lane of the backward kernels are a rule since ABI 8; ignored) */
    PBR_TUNE_BATCH_INNER = 5,           /* several lights: materials per lane of the batch-inner kernel (-1 = rule, 0 = one-material kernel) */
    PBR_TUNE_SCALAR_BASE = 6,           /* scalar plane addresses: 0 never, 1 rule (single materials), 2 whenever the launch allows them */
    PBR_TUNE_MAX_VEC = 7,               /* at most this many pixels per lane (8 default; 1 = the one-pixel kernels everywhere) */
    PBR_TUNE_BWD_RUN = 8,               /* rounds of the streamed backward kernel for fp16 maps with one light (-1 = rule, 0 = the one-tile kernels) */
    PBR_TUNE_MSE_STREAM = 9,            /* rendering-loss step for fp16 maps with one light: the streamed kernel (1, default) or the one-tile kernels (0) */
    PBR_TUNE_TILE_REPEAT = 10,          /* tiled maps: every texel loaded and decoded once and evaluated / differentiated at all its repeats (-1 = rule: on, 0 = wrap-around addressing, gradients folded by a second kernel) */
    PBR_TUNE_RESIZE_UP2 = 11,           /* the register-only resize kernels -- two taps for up-scales on both axes, the band walk for whole factors 2 ... 8 | 16 down, the row walk for other antialiased down-scales from 7 x up (1, default) -- or the strip kernels (0); 2 = as 1, with the row walk at every antialiased down-scale it can take (a test / measurement setting) */
    PBR_TUNE_COUNT = 12
};
/* (10 knobs in use since ABI 8.  ABI 6 carried 23 knobs; the 11 whose experiments are closed -- workgroup interleave, 16-byte streamed backward, the resize
 * kernels' row / tile-order / quad-store / gradient-form switches, the map kernels' launch shape, the wrap-around fold order, packed
 * one-light arithmetic, the XCD run length that pbr_render_desc.schedule already carries -- are rules now: profiles/EXPERIMENTS.md.) */
#define PBR_TUNE_UNSET INT32_MIN
#define PBR_TUNE_SLOTS 32               /* room for knobs of later versions: a pbr_tuning never changes size */
typedef struct pbr_tuning {
    int32_t knob[PBR_TUNE_SLOTS];       /* indexed by PBR_TUNE_*; PBR_TUNE_UNSET = no opinion.  Initialise with pbr_tuning_init */
} pbr_tuning;
/* Sets every entry to PBR_TUNE_UNSET. */
void pbr_tuning_init(pbr_tuning *t);

/*
 * One evaluation = CookTorranceBRDF.forward for `batch` materials and `n_lights`
 * lights.  Replaces cooktorrance.py:92-182 plus the lazy conversions it calls:
 * MaterialBase.linear_albedo (base.py:262-277), DiffuseSpecularMaterial.linear_specular
 * (diffuse.py:76-91), utils.srgb_to_linear / linear_to_srgb (utils/functions.py:31-66).
 */
typedef struct pbr_render_desc {
    int32_t abi_version;          /* PBR_HIP_ABI_VERSION */
    int32_t batch;                /* B >= 1.  The reference is unbatched (B = 1); B > 1 == a loop of calls */
    int32_t height;               /* rows of this band */
    int32_t width;                /* W */
    int32_t height_total;         /* rows of the full map: the point-light y grid spans it (cooktorrance.py:133) */
    int32_t y_offset;             /* first row of the band inside the full map (0 for a whole map) */

    int32_t map_dtype;            /* PBR_F32 | PBR_F16: albedo/normal/roughness/metallic/specular */
    int32_t out_dtype;            /* PBR_F32 | PBR_F16 */
    int32_t workflow;             /* PBR_WORKFLOW_* */
    int32_t light_type;           /* PBR_LIGHT_* */
    int32_t n_lights;             /* L in [1, PBR_MAX_LIGHTS]; L > 1: per-light clamp, sum, clamp, encode */
    int32_t albedo_is_srgb;       /* MaterialBase.albedo_is_srgb (base.py:72) */
    int32_t specular_is_srgb;     /* DiffuseSpecularMaterial.specular_is_srgb (diffuse.py:64);
                                     CONVERTED: 1 reproduces the upstream default of the converted
                                     material (already-linear specular decoded again), 0 = decoded once */
    int32_t return_srgb;          /* forward(..., return_srgb) cooktorrance.py:179-180 */

    pbr_map albedo;               /* 3 channels, required */
    pbr_map normal;               /* 3 channels, decoded to [-1,1]; data NULL = +Z (cooktorrance.py:147-152) */
    pbr_map roughness;            /* 1 channel, required */
    pbr_map metallic;             /* 1 channel: METALLIC / CONVERTED */
    pbr_map specular;             /* 3 channels: SPECULAR */
    void *out;                    /* [B][3][height][width]; contiguous unless out_*_stride (below) say otherwise */

    float view_dir[3];            /* un-normalised, as handed to forward (normalised like F.normalize, :95) */
    float light_size;             /* point lights; `light_size or 1.0` (cooktorrance.py:130) is Python truthiness: 0 (None upstream)
                                     means 1.0; a negative size mirrors the grid and NaN makes every result NaN, as upstream */
    float lights[PBR_MAX_LIGHTS][3];       /* direction (normalised here, :126) or position (:129) */
    float intensities[PBR_MAX_LIGHTS][3];  /* light_intensity per light (:96) */

    int32_t schedule;             /* workgroup -> tile order: PBR_SCHEDULE_AUTO (built-in rule), PBR_SCHEDULE_LINEAR, or
                                     PBR_SCHEDULE_XCD(c): every XCD takes runs of 1 << c consecutive tiles.  Results do
                                     not depend on it (bit-identical); pbr_cook_torrance_autotune measures the best */
    int32_t map_height;           /* MaterialBase.tile (base.py:524-537) fused as wrap-around addressing: when both are */
    int32_t map_width;            /* non-zero the maps are [B][C][map_height][map_width] and repeat over the
                                     height_total x width output (both must divide it); texel of output pixel (y, x)
                                     = ((y_offset + y) mod map_height, x mod map_width).  The point-light grid spans
                                     the OUTPUT, as it does after the reference's tile().  0/0: maps are output-sized */
    int32_t reserved;             /* 0 */
    int64_t out_batch_stride;     /* elements between the results of consecutive materials; 0 = 3 * height * width */
    int64_t out_channel_stride;   /* elements between the result's channel planes; 0 = height * width (contiguous).
                                     Rows are always contiguous.  Lets the result of material b sit right behind its
                                     maps ("material-major" batches) */
    const void *device_params;    /* NULL: view_dir / lights / intensities above are used.  Else a block of pbr_device_params_bytes() bytes
                                     written by pbr_prepare_device_params (earlier on the same stream): the kernels read view, light and
                                     intensity from it -- parameters that live in device memory (a light being fitted: cooktorrance.py:95-96,
                                     :126-140 are torch ops on device tensors upstream) never travel through the host, so a step neither
                                     synchronises nor bakes their values into a captured graph.  n_lights still counts the rows */
    const pbr_tuning *tuning;     /* NULL: the built-in rules.  Else per-call schedule knobs (above), read during the call only */
} pbr_render_desc;

#define PBR_SCHEDULE_AUTO 0
#define PBR_SCHEDULE_LINEAR 1
#define PBR_SCHEDULE_XCD(c) (1 + (c))     /* c in [1, 12] */

/* Enqueue the fused kernel.  Returns PBR_OK or an error code; never blocks. */
int pbr_cook_torrance(const pbr_render_desc *desc, void *stream);

/*
 * Which HBM channels the 8 + 3 plane streams of a launch land on depends on the buffers' addresses and
 * strides, and so does the better workgroup order (measured spread between the two orders: -6 ... +14 %).
 * Times the candidate schedules on the descriptor's own buffers (a few launches each, `out` is rewritten
 * with the same values), BLOCKS until they are done, and stores the fastest in *schedule for the caller to put
 * into desc->schedule.  Optional: PBR_SCHEDULE_AUTO picks by a rule derived from the same measurements.
 */
int pbr_cook_torrance_autotune(const pbr_render_desc *desc, void *stream, int32_t *schedule);

/*
 * Material blending fused in front of the evaluation: what examples/example_blend.py:14-32 does with
 * blend_with_mask (pypbr/blending/functional.py:64-145: every map mask*map1 + (1-mask)*map2, normals normalised,
 * blended, normalised), the re-assignment of the blended normal map (MaterialBase._process_normal_map again,
 * base.py:191-242) and CookTorranceBRDF.forward, in one pass: both materials are read once and the blended
 * maps are never written.  `desc` describes material 1 and the evaluation exactly as for pbr_cook_torrance;
 * `blend` holds material 2 (same dtype -- fp32 only --, workflow and extent; all of albedo, normal, roughness and
 * metallic|specular present in both) and the weights of material 1.  `workspace`: `batch` ints of device memory,
 * one "the blended normal map has a negative component" flag per material (base.py:212: a property of the WHOLE map).
 * sign_mode PBR_BLEND_SIGN_COMPUTE: a first small kernel on `stream` sets the flags from the maps the descriptor
 * holds -- whole maps only (a row band of an untiled map returns PBR_ERR_UNSUPPORTED).  PBR_BLEND_SIGN_GIVEN: the
 * caller has filled `workspace`, e.g. for row bands (multi-GPU sharding of one material): zero it, run
 * pbr_blend_normal_sign on every band (it only ever sets flags), combine the bands' flags (MAX), then evaluate.
 */
enum { PBR_BLEND_SIGN_COMPUTE = 0, PBR_BLEND_SIGN_GIVEN = 1 };
typedef struct pbr_blend_desc {
    pbr_map albedo, normal, roughness, metallic, specular;   /* material 2 */
    pbr_map mask;                 /* 1 channel fp32 in [0,1], [B|1][map rows][map cols]; batch_stride 0 = shared */
    int32_t sign_mode;            /* PBR_BLEND_SIGN_* */
    int32_t reserved;             /* 0 */
} pbr_blend_desc;
int pbr_blend_normal_sign(const pbr_render_desc *desc, const pbr_blend_desc *blend, void *workspace, void *stream);
int pbr_cook_torrance_blend(const pbr_render_desc *desc, const pbr_blend_desc *blend, void *workspace, void *stream);

/*
 * Gradient of pbr_cook_torrance_blend w.r.t. BOTH materials and the mask, in one pass (what autograd derives from
 * examples/example_blend.py:14-32 inside a rendering loss: CookTorranceBRDF.forward, the re-assignment of the blended normal
 * base.py:191-242, blend_with_mask blending/functional.py:64-145).  `desc` / `blend` / `workspace` exactly as for the forward
 * call (PBR_BLEND_SIGN_COMPUTE recomputes the flags; row bands need PBR_BLEND_SIGN_GIVEN); `grad_out` [B][3][H][W] fp32
 * contiguous.  Every non-NULL member of g_material1 / g_material2 and g_mask receives a contiguous fp32 gradient with one value
 * per OUTPUT pixel and material ([B][3|1][H][W]; g_mask [B][1][H][W]); a map or mask that the batch shares (batch_stride 0) owns
 * the sum over the batch, which is left to the caller (pbr_fold_gradient).
 * TILED maps (ABI 8; desc.map_height / map_width: MaterialBase.tile of the blended material, base.py:524-537 -- the blend itself is
 * map-sized, examples/example_blend.py:14-16): every gradient is MAP-sized ([B][3|1][map_height][map_width]; g_mask likewise), each texel
 * owning the sum over its repeats: one kernel blends once per texel, visits the repeats (the walk of pbr_cook_torrance_backward_folded) and
 * runs the folded gradients through the blend's chain rule.  One light, fp32, map widths of whole 4-texel groups, whole outputs or row bands
 * that hold a period of the map's rows: pbr_blend_backward_serves(desc) says whether a descriptor is served (1) or the call would return
 * PBR_ERR_UNSUPPORTED (0: evaluate the differentiable pieces -- pbr_blend_maps_backward, pbr_decode_normal_backward, the folded backward).
 */
int pbr_blend_backward_serves(const pbr_render_desc *desc);
typedef struct pbr_map_grads { void *albedo, *normal, *roughness, *metallic, *specular; } pbr_map_grads;
int pbr_cook_torrance_blend_backward(const pbr_render_desc *desc, const pbr_blend_desc *blend, void *workspace, const void *grad_out,
                                     const pbr_map_grads *g_material1, const pbr_map_grads *g_material2, void *g_mask, void *stream);

/*
 * Gradient of pbr_cook_torrance w.r.t. the maps (what torch.autograd computes through
 * cooktorrance.py:92-182 in the reference's rendering-loss use,
 * docs/source/tutorials/06_advanced.rst:73-107).  `desc` is the forward descriptor (its `out` is
 * ignored; out_dtype must be PBR_F32), fp32 or fp16 maps, any workflow (CONVERTED: gradients w.r.t. the
 * metallic-workflow maps, through the in-kernel conversion); `grad_out` is [B][3][H][W] fp32 contiguous.
 * Each non-NULL g_* receives a contiguous gradient in the maps' storage type, shaped like its map
 * ([B][3|1][H][W]); NULL skips it.  Same sub-gradient conventions as torch (clamp passes on the
 * closed interval).  With tiled maps (map_height/map_width) the g_* are OUTPUT-sized: one value per
 * output pixel; the gradient of a texel is the sum over its repeats -- pbr_cook_torrance_backward_folded hands that out directly.
 * fp16 maps with one light, rows of a whole number of 128 pixels and 4-byte-aligned planes take a streamed kernel (persistent
 * waves, the next tile prefetched global -> LDS); every other launch the one-tile kernels -- same values either way.
 */
int pbr_cook_torrance_backward(const pbr_render_desc *desc, const void *grad_out, void *g_albedo,
                               void *g_normal, void *g_roughness, void *g_metallic, void *g_specular,
                               void *stream);

/*
 * Gradient w.r.t. TILED maps, folded: MaterialBase.tile (base.py:524-537) is map.repeat(1, n, n), so autograd gives a texel the SUM of
 * the gradients of its repeats (the reference's example material is resize(512).tile(2), examples/example_brdf.py:11; its documented
 * ML use a rendering loss over such a material, docs/source/tutorials/06_advanced.rst:73-107).  Arguments as pbr_cook_torrance_backward,
 * but every non-NULL g_* is shaped like its MAP -- [B][3|1][map_height][map_width], contiguous, the maps' storage type -- and receives
 * the sum over the map's repeats inside the output the descriptor describes (the whole tiled image, or a row band of it that holds
 * at least one full period of the map's rows: a multi-GPU shard, whose partial sums the caller adds up across ranks).
 * Map rows a whole number of 4-texel groups (one or several lights): ONE kernel walks the maps, re-evaluates each texel's
 * light-independent terms once, visits its repeats and accumulates in registers (12 B per output pixel + 64 B per texel; under ONE
 * directional light, whose repeats all evaluate alike, the upstream values are summed first and the texel is differentiated once);
 * `workspace` may be NULL.  Other launches (ragged map widths) run pbr_cook_torrance_backward into `workspace`
 * (pbr_backward_folded_workspace_bytes(desc) bytes of device memory, 0 when the one-kernel form serves the descriptor) followed by
 * pbr_fold_gradient_typed per map: the same values to fp32 rounding (the one-kernel form sums a texel's adjoints over its repeats and
 * applies the light-independent tail of the chain rule once; the two-kernel form adds up per-repeat gradients); fp16 gradients are
 * rounded once instead of per repeat.
 * Untiled descriptors are passed on to pbr_cook_torrance_backward.
 */
size_t pbr_backward_folded_workspace_bytes(const pbr_render_desc *desc);
int pbr_cook_torrance_backward_folded(const pbr_render_desc *desc, const void *grad_out, void *g_albedo, void *g_normal,
                                      void *g_roughness, void *g_metallic, void *g_specular, void *workspace, void *stream);

/*
 * The same, plus the gradient w.r.t. the view / light parameters: the reference's forward is plain torch ops on
 * view_dir, light_dir_or_position and light_intensity (cooktorrance.py:95-96, :126-140), so its autograd reaches them
 * (e.g. optimising a light position against a photograph).  `g_params`: (3 + 6 L) floats of DEVICE memory, written as
 * [d/d view_dir (3) | d/d lights (L x 3) | d/d intensities (L x 3)] -- w.r.t. the values in the descriptor (view and
 * directional lights un-normalised: the F.normalize Jacobian of :95 / :126 is applied); light_size is a Python float
 * upstream and has no gradient.  `workspace`: pbr_param_grad_workspace_bytes(desc) bytes of device memory (per-workgroup
 * partial sums, added up in fp64 in a fixed order by a second small kernel on `stream`: deterministic).  Any of the
 * map gradients may be NULL.
 */
size_t pbr_param_grad_workspace_bytes(const pbr_render_desc *desc);
int pbr_cook_torrance_backward_params(const pbr_render_desc *desc, const void *grad_out, void *g_albedo,
                                      void *g_normal, void *g_roughness, void *g_metallic, void *g_specular,
                                      void *g_params, void *workspace, void *stream);

/*
 * The rendering-loss step of docs/source/tutorials/06_advanced.rst:73-107 for the PREDICTED material, as one pass:
 *     loss = nn.MSELoss()(CookTorranceBRDF(...)(predicted_material, ...), target);  loss.backward()
 * `desc` describes the predicted material and the evaluation exactly as for pbr_cook_torrance (its `out` is ignored; out_dtype
 * PBR_F32; fp32 or fp16 maps; any workflow, light type and light count); `target` is the reference rendering
 * [B][3][H][W] fp32 contiguous (e.g. pbr_cook_torrance of the ground-truth material, computed once).  Writes *loss (a DEVICE
 * float) = mean((out - target)^2) over all B*3*H*W values and, into every non-NULL g_*, d loss / d map -- contiguous, shaped
 * like the map, in the maps' storage type -- with torch's sub-gradient conventions, as pbr_cook_torrance_backward.  The
 * colour is never written: 32 + 12 bytes read and 32 written per pixel, against 44 + 36 + 76 for evaluate / MSE / backward.
 * `workspace`: pbr_mse_step_workspace_bytes(desc) bytes of device memory (one partial sum per workgroup, added in fp64 in a
 * fixed order by a second small kernel on `stream`: deterministic).  An upstream gradient other than 1 (loss * k) is applied
 * afterwards with pbr_scale_by_device_scalar, which returns at once when the scalar is 1.
 * Tiled maps (map_height / map_width, the whole output): the g_* are MAP-sized and receive the sum over the repeats, as from
 * pbr_cook_torrance_backward_folded, out of one pass over the maps and the target (12 B per output pixel + 64 B per texel); served
 * for one light and map rows of a whole number of 4-texel groups, PBR_ERR_UNSUPPORTED otherwise (pbr_mse_step_workspace_bytes
 * returns 0 then): the caller evaluates, compares and calls pbr_cook_torrance_backward_folded.
 */
size_t pbr_mse_step_workspace_bytes(const pbr_render_desc *desc);
int pbr_cook_torrance_mse_step(const pbr_render_desc *desc, const void *target, void *g_albedo, void *g_normal, void *g_roughness,
                               void *g_metallic, void *g_specular, void *loss, void *workspace, void *stream);
/* ABI 5: view / light / intensity from DEVICE memory.  `view_dir` [3], `lights` [n_lights][3], `intensities` [intensity_rows][3] with
 * intensity_rows = 1 (one intensity for every light) or n_lights: fp32 device pointers; a NULL pointer takes that parameter from the descriptor
 * (d->view_dir / d->lights / d->intensities: host values), so only what lives on the device needs to be there.  Writes `block` (pbr_device_params_bytes() bytes,
 * 16-byte aligned): the normalised view vector and per light what pbr_cook_torrance otherwise folds on the host (cooktorrance.py:95-96,
 * :126-127, :155-158).  Reads d->light_type, d->n_lights and the host parameters of the descriptor.  Put the block's address into pbr_render_desc.device_params of the
 * launches that follow on the stream (forward, backward, backward_params, blend, mse_step). */
size_t pbr_device_params_bytes(void);
int pbr_prepare_device_params(const pbr_render_desc *d, const void *view_dir, const void *lights, const void *intensities,
                              int32_t intensity_rows, void *block, void *stream);

/* data[i] *= *scalar for n elements of `dtype`, in place; `scalar` is a DEVICE float (no host synchronisation: the upstream
 * gradient of a loss lives on the device); a scalar of exactly 1 leaves the data untouched. */
int pbr_scale_by_device_scalar(void *data, size_t n, int dtype, const void *scalar, void *stream);
/* The same for up to five buffers of one dtype in ONE launch (the gradients a step leaves): data[j] has n[j] elements, count <= 5. */
int pbr_scale_list_by_device_scalar(void *const *data, const size_t *n, int count, int dtype, const void *scalar, void *stream);

/* ---- stand-alone map conversions (same arithmetic as the fused kernel) ------------- */

/* utils.srgb_to_linear, pypbr/utils/functions.py:31-47.  n elements, in-place allowed. */
int pbr_srgb_to_linear(const void *src, void *dst, size_t n, int dtype, void *stream);
/* utils.linear_to_srgb, pypbr/utils/functions.py:50-66. */
int pbr_linear_to_srgb(const void *src, void *dst, size_t n, int dtype, void *stream);

/*
 * BasecolorMetallicMaterial.to_diffuse_specular_material, metallic.py:98-108:
 * diffuse = a(1-m), specular = 0.04(1-m) + a m on LINEAR albedo (decoded here when
 * albedo_is_srgb).  albedo [B][3][P], metallic [B][1][P] -> diffuse, specular [B][3][P].
 */
int pbr_metallic_to_specular(const void *albedo, const void *metallic, void *diffuse, void *specular,
                             int32_t batch, int64_t pixels, int albedo_is_srgb, int dtype, void *stream);
/*
 * DiffuseSpecularMaterial.to_basecolor_metallic_material, diffuse.py:128-147 (raw
 * specular, 3-channel metallic).  diffuse, specular [n] -> basecolor, metallic [n].
 */
int pbr_specular_to_metallic(const void *diffuse, const void *specular, void *basecolor, void *metallic,
                             size_t n, int albedo_is_srgb, int dtype, void *stream);
/*
 * Gradients of the four conversions above w.r.t. their inputs -- what torch.autograd derives from the reference's plain torch
 * ops, so that a rendering loss differentiates through material.to_linear() / to_srgb() (base.py:754-778), the lazy
 * linear_albedo / linear_specular properties (base.py:262-277, diffuse.py:76-91), to_diffuse_specular_material (metallic.py:98-108)
 * and to_basecolor_metallic_material (diffuse.py:128-147) exactly as upstream.  torch's sub-gradient conventions: clamp passes on
 * the closed interval; masked assignment / torch.where route the gradient to the selected branch; the thresholded selects of
 * diffuse.py:136-144 are re-taken with the forward's own arithmetic.  `src` / the maps are the FORWARD INPUTS; gradients have the
 * maps' storage type and shape.  An upstream gradient that is NULL counts as zero (that output was not used); a result pointer
 * that is NULL is not computed.
 */
int pbr_srgb_to_linear_backward(const void *src, const void *grad_out, void *grad_in, size_t n, int dtype, void *stream);
int pbr_linear_to_srgb_backward(const void *src, const void *grad_out, void *grad_in, size_t n, int dtype, void *stream);
int pbr_metallic_to_specular_backward(const void *albedo, const void *metallic, const void *g_diffuse, const void *g_specular,
                                      void *g_albedo, void *g_metallic, int32_t batch, int64_t pixels, int albedo_is_srgb, int dtype,
                                      void *stream);
int pbr_specular_to_metallic_backward(const void *diffuse, const void *specular, const void *g_basecolor, const void *g_metallic,
                                      void *g_diffuse, void *g_specular, size_t n, int albedo_is_srgb, int dtype, void *stream);

/* Gradient of pbr_decode_normal w.r.t. the stored map (what autograd computes through base.py:191-242 when a predicted
 * normal map is assigned to a material in a rendering loss): fp32, `workspace` = the flag pbr_decode_normal left. */
int pbr_decode_normal_backward(const void *src, const void *grad_out, void *grad_in, int32_t channels, int64_t pixels,
                               const void *workspace, void *stream);

/* Gradient folding behind pbr_cook_torrance_backward (the sums torch.autograd would perform for a broadcast or a
 * repeat(): a map shared by the whole batch, or tiled ny x nx by map_height/map_width, owns the sum of the
 * per-output-pixel gradients).  src [batch][channels][ny*h][nx*w] fp32 contiguous ->
 * dst [fold_batch ? 1 : batch][channels][h][w]. */
int pbr_fold_gradient(const void *src, void *dst, int32_t batch, int32_t channels, int32_t h, int32_t w, int32_t ny,
                      int32_t nx, int fold_batch, void *stream);
/* The same for gradients stored as the maps are (pbr_cook_torrance_backward returns fp16 gradients for fp16 maps): `dtype`
 * PBR_F32 | PBR_F16 for src and dst alike, the sums are formed in fp32 and rounded once. */
int pbr_fold_gradient_typed(const void *src, void *dst, int32_t batch, int32_t channels, int32_t h, int32_t w, int32_t ny,
                            int32_t nx, int fold_batch, int dtype, void *stream);

/*
 * MaterialBase._process_normal_map, base.py:191-242.  channels = 2 or 3, planar
 * [channels][pixels] -> [3][pixels].  3 channels: kept as-is when any value is
 * negative, else x*2-1 and unit length.  `workspace` = 4 bytes of device memory
 * (the min<0 flag; written by the call: 1 when the map was kept as it is).
 * Source and destination disjoint: the map is read once (twice when it is kept
 * as it is only because of a negative value the 4096-sample probe missed).
 * `dst == src` is allowed: flag pass first, then the transform.
 */
int pbr_decode_normal(const void *src, void *dst, int32_t channels, int64_t pixels, int dtype,
                      void *workspace, void *stream);

/*
 * MaterialBase._to_tensor for PIL images, base.py:143-164, on the device: an image's own samples -- uint8 (`bits` 8; torchvision's
 * to_tensor: (H,W,C) -> float32 (C,H,W) / 255) or uint16 (`bits` 16; base.py:146-152: / 65535.0) -- become the float32 planar map
 * dst [channels][height][width] (dense).  The division is IEEE-exact: every one of the 256 / 65 536 possible samples gives the float
 * the reference's CPU code gives.  `src` is addressed as src[c*stride_c + y*stride_h + x*stride_w] (strides in samples), so the
 * (H,W,C) array PIL hands out travels as it is -- a quarter of the bytes of the float map on the host-to-device copy -- and needs
 * no transposing on the host; channels 1..4.
 * decode_normal != 0: the map is a normal map (channels 2 or 3) and base.py:191-242 `_process_normal_map` follows in the same pass --
 * samples are never negative, so :212's "already signed?" is false by construction and the map is always decoded; dst has 3 planes,
 * bit-identical to pbr_decode_normal of the float map.
 */
int pbr_unpack_image(const void *src, int32_t bits, int32_t channels, int32_t height, int32_t width, int64_t stride_c,
                     int64_t stride_h, int64_t stride_w, float *dst, int32_t decode_normal, void *stream);

/*
 * MaterialBase.resize, base.py:490-504 (torchvision resize of a float (C,H,W) map ==
 * F.interpolate(mode="bilinear", align_corners=False, antialias=...)).  fp32 planar
 * [planes][h_in][w_in] -> [planes][h_out][w_out]; `workspace` holds the width-pass
 * intermediate (or the row walk's tap tables), pbr_resize_workspace_bytes(planes, h_in, w_out) bytes of device memory, 16-byte aligned.
 */
size_t pbr_resize_workspace_bytes(int64_t planes, int32_t h_in, int32_t w_out);
int pbr_resize_bilinear(const void *src, void *dst, int64_t planes, int32_t h_in, int32_t w_in,
                        int32_t h_out, int32_t w_out, int antialias, void *workspace, void *stream);
/* Which kernel family pbr_resize_bilinear runs for these arguments under the current knobs (nothing is launched; -1 = the call would
 * fail): what a caller's test asserts when it compares two families bit for bit, what a profile's reader looks up.  ABI 8. */
enum {
    PBR_RESIZE_TWO_PASS = 0,            /* width pass, height pass through the workspace (more than 36 taps per axis) */
    PBR_RESIZE_STRIP = 1,               /* one kernel, tile by tile through an LDS strip (csrc/resize.hip: resize_strip_kernel) */
    PBR_RESIZE_TWO_TAP = 2,             /* up-scales on both axes, registers only (resize_up2_kernel) */
    PBR_RESIZE_BAND_WALK = 3,           /* antialiased whole factors 2 ... 8 | 16, registers only (csrc/resize_down.hpp) */
    PBR_RESIZE_ROW_WALK = 4             /* other antialiased down-scales from 7 x up, i.e. 17 ... 36 taps per axis (every one from 1.01 x to 17 x with the knob PBR_TUNE_RESIZE_UP2 at 2): every input row read once (csrc/resize_stream.hpp) */
};
int pbr_resize_form(const void *src, const void *dst, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                    int32_t w_out, int antialias, const void *workspace);

/*
 * Gradient of pbr_resize_bilinear w.r.t. its input (autograd through base.py:490-504 -> F.interpolate): the transposed tap
 * matrices, g_in = Wy^T g_out Wx, gathered by input index in a fixed order (deterministic, no atomics).  grad_out
 * [planes][h_out][w_out] -> grad_in [planes][h_in][w_in], fp32; `workspace`: pbr_resize_backward_workspace_bytes(...) bytes.
 */
size_t pbr_resize_backward_workspace_bytes(int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out, int32_t w_out);
int pbr_resize_bilinear_backward(const void *grad_out, void *grad_in, int64_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                                 int32_t w_out, int antialias, void *workspace, void *stream);

/*
 * Material blending in front of the BRDF (examples/example_blend.py:14-16), fp32 planar maps.
 * pbr_blend_maps: blend_with_mask blending/functional.py:64-116 for ONE map, out = mask * map1 +
 * (1 - mask) * map2 over [channels][pixels] with a [pixels] mask; is_normal selects _blend_normals
 * (:119-145: normalise both, blend, re-normalise; 3 channels).
 * pbr_blend_sigmoid_mask: the mask of blend_on_height / blend_on_properties (:148-239),
 * sigmoid((prop1 + shift - prop2) / (blend_width + 1e-6)).
 * pbr_blend_gradient_mask: the mask of blend_with_gradient (:242-286), linspace(0,1) along x
 * (vertical = 0) or y (vertical = 1), shape [height][width].
 */
int pbr_blend_maps(const void *map1, const void *map2, const void *mask, void *out, int32_t channels,
                   int64_t pixels, int is_normal, void *stream);
/* Gradient of pbr_blend_maps (autograd through functional.py:103-110 / :119-145): g_map1, g_map2 [channels][pixels] and
 * g_mask [pixels], any of them NULL = not wanted; accumulate_mask: g_mask += (a material's maps share one mask). */
int pbr_blend_maps_backward(const void *map1, const void *map2, const void *mask, const void *grad_out, void *g_map1,
                            void *g_map2, void *g_mask, int32_t channels, int64_t pixels, int is_normal,
                            int accumulate_mask, void *stream);
int pbr_blend_sigmoid_mask(const void *prop1, const void *prop2, void *mask, int64_t n, float shift,
                           float blend_width, void *stream);
/* Gradient of pbr_blend_sigmoid_mask w.r.t. both property maps (autograd through torch.sigmoid, functional.py:184-193), from the
 * mask the forward call produced: g_prop1 = grad_out * mask (1 - mask) / (blend_width + 1e-6), g_prop2 = -g_prop1; NULL = not wanted. */
int pbr_blend_sigmoid_mask_backward(const void *mask, const void *grad_out, void *g_prop1, void *g_prop2, int64_t n, float blend_width,
                                    void *stream);
int pbr_blend_gradient_mask(void *mask, int32_t height, int32_t width, int vertical, void *stream);

/* ---- introspection; the process-global tuning hook (tests, benches, profiling) ------ */
int pbr_abi_version(void);
/* First 16 hex digits of the SHA-256 over the sources this library was built from (every .hip and .hpp file of csrc, this header, the Makefile,
 * in sorted order): evidence files carry it, and collectors refuse a library that is not the one the sources next to it would build. */
const char *pbr_build_id(void);
/* sizeof(pbr_render_desc) as compiled: bindings check their struct layout against it. */
size_t pbr_render_desc_size(void);
const char *pbr_error_string(int code);
/* Name of the kernel the descriptor dispatches to (no launch); NULL on a bad descriptor. */
const char *pbr_kernel_name(const pbr_render_desc *desc);
/* Algorithmic HBM bytes per pixel of that dispatch: reads + writes. */
int pbr_bytes_per_pixel(const pbr_render_desc *desc);
/* The process-global hook behind the knobs above (pbr_tuning): sets the value every later call of the PROCESS uses for `knob`
 * unless its descriptor overrides it; returns the previous value, -1 for an unknown knob.  For tests, benches and profiling runs
 * (the Python binding feeds PBR_TUNE_* environment variables through it at load).  Atomic, so it may be called while other threads
 * launch; it is still shared state -- a knob left set changes which kernel a later, unrelated caller gets (never its results) --
 * which is why product code uses pbr_render_desc.tuning.  pbr_set_tuning(knob, PBR_TUNE_UNSET) restores the rule. */
int pbr_set_tuning(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* PBR_HIP_H */
