/*
 * tiatoolbox_amd -- C ABI of the MI355X (gfx950) hot-path library  (libtiatoolbox_amd.so)
 *
 * Every entry point takes raw DEVICE pointers + sizes + a hipStream_t (passed as void*).
 * Naming convention: `d_*` arguments are device pointers (data planes, workspaces, outputs);
 * small parameter blocks that are read on the host while the call is being enqueued are
 * passed by `const` pointer WITHOUT the prefix (`params`, `h_target_stain`, `target_means`,
 * `target_stds`; each is marked "host" where it is declared) -- they play the role of
 * by-value arguments and are copied at launch.
 * No allocation, no synchronisation, no exceptions cross this boundary; each call
 * enqueues kernels on `stream` and returns 0 (TIA_OK) or a negative TIA_E* code.
 *
 * The reference (tiatoolbox v2.0.1) is pure Python; the "FFI" a maintainer would add is
 * a ctypes binding.  Each entry point cites the reference call
 * site(s) it replaces as `file:line` relative to /root/reference/tiatoolbox/.
 */
#ifndef TIATOOLBOX_AMD_H
#define TIATOOLBOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIA_OK 0
#define TIA_EINVAL (-1)   /* bad argument (null pointer, non-positive size, bad mode)   */
#define TIA_ELAUNCH (-2)  /* hipLaunch / runtime failure (hipGetLastError() != success)  */
#define TIA_ESIZE (-3)    /* size outside what the kernel supports                        */

/* Version 3 (round 3): + tia_stem_pack_weights_f32 / tia_stem_conv7x7_pool_nhwc, tia_conv_pack_weights_h / tia_conv2d_nhwc_h,
 * tia_stem_pack_weights_h / tia_stem_conv7x7_pool_nhwc_h, tia_conv2d_thin_nhwc_f32, tia_conv1x1_head_nhwc_f32, tia_lut_apply_u8, tia_box_downsample_u8; the workspace of
 * tia_stain_stats_u8 grew by one int32 flag per patch (tia_stain_stats_workspace_bytes_mode reports it). */
/* Version 4 (round 4): + tia_rgb2od_u8 (the stand-alone OD transform, with the reference's in-place side effect on request),
 * tia_clear_last_error; TIA_MATH_F64 of tia_stain_apply_u8 evaluates exp() with the library's own float64 kernel
 * (TIA_MATH_F64_REF keeps the device libm's exp); tia_conv3x3_geometry, tia_stain_stats_path (dispatch diagnostics). */
/* Version 6 (round 6): + tia_gray_hist_u8 / tia_otsu_threshold_u32 / tia_otsu_fit_u8 / tia_threshold_lt_dev_u8 (one-pass, one-launch
 * Otsu fit whose threshold stays on the device), tia_morph_mask_u8 (the morphological masker in one launch), tia_reinhard_transform_u8 /
 * tia_reinhard_workspace_bytes / tia_lab_moments_u8 (one-launch Reinhard); tia_luminosity_mask_u8 and the float64 form of
 * tia_stain_augment_u8 (now a product of per-patch tables) take 16-byte accesses where the shape allows. */
#define TIA_ABI_VERSION 6
int tia_abi_version(void);

/* Reads-and-clears the HIP runtime's sticky last-error value as THIS library sees it (every entry point returns
 * `hipGetLastError() == hipSuccess ? TIA_OK : TIA_ELAUNCH`).  Call it after a host-side HIP call that is allowed to be
 * refused (hipHostRegister of already-pinned memory) so that the refusal does not surface as a later TIA_ELAUNCH.
 * Returns the value that was pending (0 = none). */
int tia_clear_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Stain tables (device buffer, built once by the host; layout = struct tia_stain_tables).
 * od_lut[v]      = max(-ln(max(v,1)/255), 1e-6)            utils/transforms.py:229-231
 * ty[c][v]       = C[3+c] * sRGBGammaTab_b[v]   (Y row of OpenCV's 8-bit RGB2Lab)
 *                  -> L channel for get_luminosity_tissue_mask, utils/misc.py:281-283
 * ------------------------------------------------------------------------------------- */
typedef struct tia_stain_tables {
    double od_lut[256];
    float od_lut_f32[256];
    int32_t ty[3][256];
} tia_stain_tables;

/* Per-patch statistics record: TIA_STATS_STRIDE doubles per patch (512 B, cache-line aligned). */
#define TIA_STATS_STRIDE 64
#define TIA_ST_STAIN 0    /* [6]  source stain matrix, rows H,E unit norm  (stainextract.py:177-227) */
#define TIA_ST_MAXC 6     /* [2]  99th pct of source concentrations        (stainnorm.py:103)        */
#define TIA_ST_NTISSUE 8  /*      tissue pixel count                                                   */
#define TIA_ST_PLOW 9     /*      contrast-enhancer p_low                   (utils/misc.py:434-437)   */
#define TIA_ST_PHIGH 10   /*      contrast-enhancer p_high                                             */
#define TIA_ST_MINPHI 11  /*      percentile(phi, 100-ap)                   (stainextract.py:217)     */
#define TIA_ST_MAXPHI 12  /*      percentile(phi, ap)                       (stainextract.py:218)     */
#define TIA_ST_COV 13     /* [6]  OD covariance xx,xy,xz,yy,yz,zz (ddof=1)  (stainextract.py:202)     */
#define TIA_ST_EVEC 19    /* [6]  principal eigenvectors v1[3], v2[3] after sign fix (:205-208)       */
#define TIA_ST_FLAGS 25   /*      bit0: empty tissue mask; bit1: degenerate (n<2 / non-finite)        */
#define TIA_ST_PINV 26    /* [6]  pinv(S^T) as P[j*2+i], C_i = sum_j OD_j P[j][i] (stainnorm.py:65)   */
#define TIA_ST_M 32       /* [9]  M[j*3+c] = sum_i P[j][i]*(maxC_t[i]/maxC_s[i])*S_t[i][c]            */
#define TIA_ST_SCALE 41   /* [2]  maxC_target / maxC_source                 (stainnorm.py:104)        */
#define TIA_ST_CYCLES 48  /* [16] shader-clock cycles per phase (instrumentation: -DTIA_STATS_TIMING=1 builds of stain_stats_*.hip) */

#define TIA_FLAG_EMPTY_MASK 1
#define TIA_FLAG_DEGENERATE 2

#define TIA_MODE_MACENKO 0 /* estimate the stain matrix per patch (MacenkoExtractor)       */
#define TIA_MODE_FIXED 1   /* stain matrix given (Custom / Ruifrok)                         */
#define TIA_MODE_GIVEN 3   /* as TIA_MODE_FIXED, but every patch brings its own stain matrix: d_stats[i][TIA_ST_STAIN..+5]
                              is read on entry (StainAugmentor.fit on a batch, stainaugment.py:141-175) */
#define TIA_MODE_VAHADANE 2 /* per-patch dictionary learning (VahadaneExtractor, stainextract.py:281-322): X = tissue
                              OD transposed (3 x N), sklearn DictionaryLearning(n_components=2, alpha, fit_algorithm="lars",
                              positive_dict=True, max_iter) restated -- SVD initialisation, two-atom lasso codes, in-place
                              dictionary update with the 1e-6 unused-atom rule, cost-based stopping; the stain matrix is
                              the code transposed, H row first, unit rows */

typedef struct tia_stain_params {
    double q_img_lo;        /* contrast_enhancer low percentile / 100   (0.02) */
    double q_img_hi;        /* contrast_enhancer high percentile / 100  (0.98) */
    double q_phi_lo;        /* (100 - angular_percentile) / 100          (0.01) */
    double q_phi_hi;        /* angular_percentile / 100                  (0.99) */
    double q_conc;          /* concentration percentile / 100            (0.99) */
    double stain_fixed[6];  /* TIA_MODE_FIXED: the 2x3 stain matrix                          */
    double target_stain[6]; /* fitted target stain matrix (for TIA_ST_M); ignored if !has_target */
    double target_maxc[2];  /* fitted target maxC                                            */
    int32_t y_thr;          /* tissue <=> descaled Y index < y_thr (from luminosity threshold) */
    int32_t mode;           /* TIA_MODE_*                                                    */
    int32_t has_target;     /* 1: also emit TIA_ST_M / TIA_ST_SCALE                          */
    int32_t zero_to_one;    /* 1: treat byte 0 as 1 in the mask path (rgb2od's in-place edit
                               seen by StainAugmentor.fit, stainaugment.py:163-175)          */
    double dl_alpha;        /* TIA_MODE_VAHADANE: regularizer (alpha = transform_alpha, 0.1; stainextract.py:307-308) */
    double dl_tol;          /* DictionaryLearning.tol (1e-8)                                  */
    int32_t dl_max_iter;    /* DictionaryLearning.max_iter (3; stainextract.py:313)           */
    int32_t dl_seed;        /* stream of the unused-atom re-draw (the reference is unseeded)  */
    int32_t select_mode;    /* order statistics: 0 = sample-placed windows (one float32 sweep + exact candidates) with the
                               histogram path as fall-back, on the register-resident kernel where the patch fits;
                               1 = histogram path only; 2 = windows on the streaming kernel only (same results bit for bit
                               in all three; 1 and 2 are parity audits) */
    int32_t dl_one_kernel;  /* TIA_MODE_VAHADANE, parity audit: 0 = the kernel pair (dictionary learning by per-pixel replay of
                               the atom updates -- no dictionary in memory -- followed by the common tail); 1 = the one-kernel form
                               that keeps the 2 x N float64 dictionary in the workspace.  Same results bit for bit.  (Was
                               `reserved`, 0, before ABI version 4.) */
} tia_stain_params;

/*
 * Per-patch stain statistics for a batch of NHWC uint8 patches (one workgroup per patch).
 * Replaces, per patch: MacenkoExtractor.get_stain_matrix (tools/stainextract.py:177-227,
 * incl. utils/misc.py:261-290,405-444 and utils/transforms.py:209-231), the lstsq of
 * StainNormalizer.get_concentrations (tools/stainnorm.py:49-66) as a closed-form
 * pseudo-inverse, and np.percentile(C, 99, axis=0) (tools/stainnorm.py:81-85,103).
 *   d_img    [n,h,w,3] uint8     d_tables  tia_stain_tables     d_stats [n,TIA_STATS_STRIDE] f64
 *   d_ws     optional scratch of tia_stain_stats_workspace_bytes(n,h,w) bytes (8-byte aligned): a
 *            per-pixel histogram-bin cache that lets the selection's collect passes skip the value
 *            computation; with NULL / too little space the kernel recomputes (same results).
 *            TIA_MODE_VAHADANE needs tia_stain_stats_workspace_bytes_mode(n,h,w,mode) bytes, 16-byte aligned
 *            (bin cache + the 2 x N float64 dictionary of every patch); too little -> TIA_ESIZE.
 */
size_t tia_stain_stats_workspace_bytes(int64_t n, int64_t h, int64_t w);
size_t tia_stain_stats_workspace_bytes_mode(int64_t n, int64_t h, int64_t w, int32_t mode);
int tia_stain_stats_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                       const tia_stain_tables* d_tables, const tia_stain_params* params,
                       double* d_stats, void* d_ws, size_t ws_bytes, void* stream);
/* Which of the two statistics kernels serves patches of h x w with these parameters, given an aligned, full-size workspace
 * (diagnostics for tests and profiles; host only): 0 = the streaming kernel (the patch is re-read per sweep), 1 = the
 * register-resident kernel (one read; patches it hands back go through the streaming kernel).  Same bits either way. */
int tia_stain_stats_path(int64_t h, int64_t w, const tia_stain_params* params /* host */);

/* Output kinds of tia_stain_apply_u8 */
#define TIA_OUT_U8 0       /* uint8 NHWC, astype(uint8) truncation   (stainnorm.py:110-113)          */
#define TIA_OUT_F32 1      /* float32 NHWC: the pre-cast float in [0,255] (parity instrumentation)   */
#define TIA_OUT_F64 2      /* float64 NHWC: idem, f64                                                  */
#define TIA_OUT_UNIT_F16 3 /* half NHWC = uint8(trunc)/255 : ToTensor() of the normalised patch       */
#define TIA_OUT_UNIT_BF16 4
#define TIA_OUT_UNIT_F32 5

#define TIA_MATH_F64 0 /* float64 throughout, 3x3 matrix TIA_ST_M fused in float64; 255 exp(-OD') evaluated as the
                          product of three per-patch table entries T_j[c][v] = exp(-LUT[v] m[j][c]) built with libm
                          (16-byte-access kernels, round 5; relative error <= ~1e-15) or by the kernel's own table +
                          cubic (12-byte-access kernels; <= 4e-15): < 1e-12 on the 0..255 scale either way */
#define TIA_MATH_F32 1 /* fused 3x3 matrix in f32: |err| <= 1e-4 on the pre-cast float            */
#define TIA_MATH_F64_REF 2 /* the reference's order of operations in f64 (stainnorm.py:102-107) with the device
                              library's exp(): parity audit of TIA_MATH_F64, and its in-kernel fall-back for patches
                              whose matrix is not finite / outside the table arithmetic's safe range    */

/*
 * out = 255*exp(-(OD(img) . pinv . diag(scale) . S_target)), clipped to [0,255].
 * Streaming kernel: 1 read + 1 write per pixel.  Replaces tools/stainnorm.py:102-113.
 * Reads TIA_ST_M / TIA_ST_PINV / TIA_ST_SCALE from d_stats (produced with has_target=1).
 */
int tia_stain_apply_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                       const tia_stain_tables* d_tables, const double* d_stats,
                       const double* h_target_stain /* host parameter block, double[6] */, void* d_out, int32_t out_kind,
                       int32_t math, void* stream);

/*
 * Concentrations C[n*h*w, 2] (f64) of every pixel w.r.t. the per-patch TIA_ST_PINV.
 * Replaces StainNormalizer.get_concentrations (tools/stainnorm.py:49-66).
 */
int tia_stain_concentrations_f64(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                                 const tia_stain_tables* d_tables, const double* d_stats,
                                 double* d_conc, void* stream);

/*
 * StainAugmentor.augment (tools/stainaugment.py:177-206): C[mask,i] = C[mask,i]*alpha[i]+beta[i]
 * (all pixels if augment_background), out = uint8(clip(255*exp(-C.S),0,255)).  S = per-patch
 * TIA_ST_STAIN, mask from TIA_ST_PLOW/PHIGH + tables.  d_alpha_beta: [n,4] f64 = a0,a1,b0,b1.
 * math = TIA_MATH_F64: float64 -- for whole 3072-byte chunks and 16-byte aligned buffers as a product of per-patch tables (the
 * augmented optical density is affine in the three input optical densities: two sets of nine 256-entry tables replace three
 * exponentials per pixel; < 1e-13 from the per-pixel arithmetic on the 0..255 scale), else / for degenerate stain matrices the
 * reference's per-pixel arithmetic with the device libm's exp; TIA_MATH_F32: float32 with hardware exp2 and
 * 16-byte global accesses (needs h*w*3 % 3072 == 0 and 16-byte aligned buffers, else TIA_ESIZE).
 */
int tia_stain_augment_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                         const tia_stain_tables* d_tables, const double* d_stats,
                         const double* d_alpha_beta, int32_t y_thr, int32_t augment_background,
                         int32_t zero_to_one, uint8_t* d_out, int32_t math, void* stream);

/* Luminosity tissue mask (utils/misc.py:261-290) as uint8 0/1 [n,h,w]; needs TIA_ST_PLOW/PHIGH. */
int tia_luminosity_mask_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w,
                           const tia_stain_tables* d_tables, const double* d_stats, int32_t y_thr,
                           int32_t zero_to_one, uint8_t* d_mask, void* stream);

/*
 * rgb2od (utils/transforms.py:209-231) of `nbytes` uint8 values of any shape: d_od[i] = max(-ln(max(d_img[i],1)/255), 1e-6)
 * as float64 (table look-up: the 256 possible results, computed on the host with NumPy's own log).  mutate = 1 also
 * performs the reference's side effect on its argument, `img[img == 0] = 1` (:229-230), in place on d_img.
 */
int tia_rgb2od_u8(uint8_t* d_img, int64_t nbytes, const tia_stain_tables* d_tables, int32_t mutate, double* d_od, void* stream);


/* =======================================================================================
 * Image primitives shared by the tissue maskers (tools/tissuemask.py) and the HoVer-Net
 * post-processing (models/architecture/hovernet.py:502-616).  All images are [n,h,w] planes.
 * ===================================================================================== */

/* cv2.cvtColor(COLOR_RGB2GRAY) 8-bit: (R*9798 + G*19235 + B*3735 + 2^14) >> 15
 * (tools/tissuemask.py:129,160,291).  d_img [npix,3] -> d_gray [npix]. */
int tia_rgb2gray_u8(const uint8_t* d_img, int64_t npix, uint8_t* d_gray, void* stream);

/* 256-bin histogram of bytes, ACCUMULATED into d_hist[256] (zero it first); feeds
 * skimage.filters.threshold_otsu (tools/tissuemask.py:131-134). */
int tia_hist256_u8(const uint8_t* d_data, int64_t n, uint32_t* d_hist, void* stream);

/* mask = (gray < thr) as 0/1 bytes; with is_rgb the grey conversion is fused
 * (tools/tissuemask.py:156-162, 288-296). */
int tia_threshold_lt_u8(const uint8_t* d_src, int64_t npix, int32_t is_rgb, int32_t thr,
                        uint8_t* d_mask, void* stream);

/* OtsuTissueMasker.fit in one pass (tools/tissuemask.py:127-137): the grey conversion of tia_rgb2gray_u8 (channels == 3; a grey
 * plane with channels == 1) fused with the 256-bin histogram, ACCUMULATED into d_hist[256] (zero it first; several calls add
 * several images).  Nothing is written but the counts: 3 bytes read per pixel. */
int tia_gray_hist_u8(const uint8_t* d_img, int64_t npix, int32_t channels, uint32_t* d_hist, void* stream);

/* skimage.filters.threshold_otsu of a 256-bin byte histogram on the device (tools/tissuemask.py:131-134): d_out[0] = threshold
 * (bin centre of the first maximum of the between-class variance over the occupied range; the only occupied bin when there is one),
 * d_out[1] = number of occupied bins.  Bit-identical to the float64 NumPy arithmetic (all partial sums are exact integers). */
int tia_otsu_threshold_u32(const uint32_t* d_hist, int32_t* d_out, void* stream);

/* OtsuTissueMasker.fit in ONE launch: tia_gray_hist_u8 whose last-finishing workgroup runs tia_otsu_threshold_u32's arithmetic on
 * the completed counts.  d_hist: 257 uint32 -- the 256 bins and a ticket counter -- zeroed by the caller before the first image
 * (the kernel leaves the counter at zero, so a later call that adds another image to the same counts recomputes d_out). */
int tia_otsu_fit_u8(const uint8_t* d_img, int64_t npix, int32_t channels, uint32_t* d_hist, int32_t* d_out, void* stream);

/* tia_threshold_lt_u8 with the threshold read from device memory (d_thr[0], e.g. tia_otsu_threshold_u32's output): fit and
 * transform enqueue back to back without a host round trip (tools/tissuemask.py:139-164). */
int tia_threshold_lt_dev_u8(const uint8_t* d_src, int64_t npix, int32_t is_rgb, const int32_t* d_thr, uint8_t* d_mask,
                            void* stream);

/* MorphologicalMasker.transform in one launch (tools/tissuemask.py:270-306): grey < thr (thr from d_thr[0] when d_thr is given),
 * skimage remove_small_objects(min_size = min_region, connectivity 8), cv2.dilate with the element whose non-zero entries are
 * d_offsets [n_off,2] = (dy, dx) relative to the anchor (`reach` = max |dy|, |dx|).  d_img [n,h,w,channels] (3: RGB, the grey
 * conversion fused; 1: grey plane) -> d_mask [n,h,w] 0/1.  Tiles of 256 x 128 pixels with a halo of reach + min_region - 1 are
 * labelled in LDS; a halo above 40 pixels returns TIA_ESIZE (use tia_threshold_lt_u8 + tia_ccl_label_i32 +
 * tia_label_area_filter_i32 + tia_binary_morph_u8). */
int tia_morph_mask_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, int32_t channels, int32_t thr, const int32_t* d_thr,
                      int32_t min_region, const int32_t* d_offsets, int32_t n_off, int32_t reach, uint8_t* d_mask, void* stream);

/* Connected-component labelling of n binary planes (non-zero = foreground), connectivity 4 or 8.
 * Labels are 1..K per plane, numbered in raster order of each component's first pixel
 * (= scipy.ndimage.label, hovernet.py:543,607; cv2.connectedComponentsWithStats labelling,
 * tissuemask.py:297).  d_labels [n,h,w] i32, d_count [n] i32, d_ws: n*h*w i32 scratch. */
int tia_ccl_label_i32(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, int32_t connectivity,
                      int32_t* d_labels, int32_t* d_count, int32_t* d_ws, void* stream);

/* Zero every label whose pixel count is < min_keep (no relabelling):
 * skimage remove_small_objects(max_size=s) == min_keep = s+1 (hovernet.py:544,614);
 * MorphologicalMasker min_region_size == min_keep (tissuemask.py:297-301).
 * d_ws: n*(h*w+1) i32 scratch (per-label areas). */
int tia_label_area_filter_i32(int32_t* d_labels, int64_t n, int64_t h, int64_t w, int32_t min_keep,
                              int32_t* d_ws, void* stream);

/* Binary morphology with an arbitrary structuring element given as n_off (dy,dx) int32 pairs
 * relative to the anchor.  op 0 = dilate (outside = 0), 1 = erode (outside = 1), i.e. OpenCV's
 * morphologyEx with the default border value (tissuemask.py:303; hovernet.py:605-606). */
int tia_binary_morph_u8(const uint8_t* d_src, int64_t n, int64_t h, int64_t w, const int32_t* d_offsets,
                        int32_t n_off, int32_t op, uint8_t* d_dst, void* stream);

/* scipy.ndimage.binary_fill_holes with the default (4-connected) structure (hovernet.py:604):
 * background components not connected to the border become foreground.
 * d_ws: (2*n*h*w + n) i32 scratch. */
int tia_fill_holes_u8(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, uint8_t* d_out,
                      int32_t* d_ws, void* stream);


/* =======================================================================================
 * HoVer-Net post-processing (models/architecture/hovernet.py:502-748)
 * ===================================================================================== */

/* Bytes of scratch tia_hover_proc_np_hv_f32 needs for n planes of h x w. */
size_t tia_hover_workspace_bytes(int64_t n, int64_t h, int64_t w);

/*
 * HoVerNet._proc_np_hv (hovernet.py:502-616) for n patches/tiles at once:
 * binarise np>=0.5 -> 4-connected labelling -> drop objects <= 9 px -> min-max normalise the
 * h/v maps (f32) -> Sobel ksize (CV_64F, separable, REFLECT_101) -> normalise (f32) ->
 * energy / distance maps (f64) -> 3x3 Gaussian -> marker = blb - (overall>=0.4), fill holes,
 * 5x5 elliptical opening, labelling, drop objects < obj_size -> marker-controlled watershed
 * (priority flood, one independent flood per mask blob).
 *   d_np [n,h,w] f32   d_hv [n,h,w,2] f32   d_inst [n,h,w] i32 (instance ids = marker ids)
 *   d_ninst [n] i32 (number of marker labels = max possible id)
 */
int tia_hover_proc_np_hv_f32(const float* d_np, const float* d_hv, int64_t n, int64_t h, int64_t w,
                             int32_t ksize, int32_t obj_size, int32_t* d_inst, int32_t* d_ninst,
                             void* d_ws, size_t ws_bytes, void* stream);

/*
 * Same pipeline with optional copies of its intermediate planes (any of the five may be NULL), so each stage can be
 * compared with the reference's corresponding local (hovernet.py:547-614) instead of only the final label map:
 *   d_sobel_h / d_sobel_v [n,h,w] f64 = cv2.Sobel(normalised h|v, CV_64F, ksize) before the second normalisation (:554-555)
 *   d_dist [n,h,w] f64 = -cv2.GaussianBlur((1 - overall) * blb, (3,3), 0)                                      (:599-600)
 *   d_markers [n,h,w] i32 = labelled, size-filtered markers fed to the watershed                             (:607-614)
 *   d_blobs [n,h,w] i32 = blb (0/1) after remove_small_objects                                               (:543-545)
 */
int tia_hover_proc_np_hv_stages_f32(const float* d_np, const float* d_hv, int64_t n, int64_t h, int64_t w,
                                    int32_t ksize, int32_t obj_size, int32_t* d_inst, int32_t* d_ninst,
                                    double* d_sobel_h, double* d_sobel_v, double* d_dist, int32_t* d_markers,
                                    int32_t* d_blobs, void* d_ws, size_t ws_bytes, void* stream);

/* Bytes of scratch tia_watershed_blobs_f64 needs for n planes of h x w. */
size_t tia_watershed_workspace_bytes(int64_t n, int64_t h, int64_t w);

/*
 * skimage.segmentation.watershed(image, markers, mask=mask) with connectivity 1, no compactness, no watershed line
 * (the call at hovernet.py:616), for n planes: priority flood ordered by (value, insertion age) with skimage's own
 * binary-heap procedures (heap_general.pxi), neighbours visited up / left / right / down, a pixel labelled when it is
 * pushed.  Floods of different 4-connected mask components are independent and run concurrently, one wave each.
 *   d_image [n,h,w] f64   d_markers [n,h,w] i32 (>= 0; 0 = no marker)   d_mask [n,h,w] u8 (non-zero = flood here)
 *   d_out [n,h,w] i32: label of the marker whose flood reached the pixel; 0 outside the mask and in unreached blobs
 */
int tia_watershed_blobs_f64(const double* d_image, const int32_t* d_markers, const uint8_t* d_mask, int64_t n,
                            int64_t h, int64_t w, int32_t* d_out, void* d_ws, size_t ws_bytes, void* stream);

/*
 * Per-instance statistics for HoVerNet.get_instance_info (hovernet.py:670-748): for every id in
 * 1..max_inst of every plane: pixel count, bounding box (x_min,y_min,x_max,y_max inclusive),
 * sum of x, sum of y, and the histogram of d_type (uint8, may be NULL) over its pixels.
 *   d_stats [n, max_inst+1, 8] i64 = area, xmin, ymin, xmax, ymax, sumx, sumy, 0
 *   d_types [n, max_inst+1, num_types] i32
 * Both must be zero-initialised except xmin/ymin which the call initialises itself.
 */
int tia_hover_instance_stats(const int32_t* d_inst, const uint8_t* d_type, int64_t n, int64_t h, int64_t w,
                             int32_t max_inst, int32_t num_types, int64_t* d_stats, int32_t* d_types,
                             void* stream);

/*
 * Contour polygon of every instance: cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]
 * of the instance's binary mask (hovernet.py:685-692) = the top-level outer border found last in
 * raster order, vertices where the 8-connected chain code changes, counter-clockwise on screen
 * starting at the border's first pixel in raster order.  One lane follows the borders of one instance.
 *   scan : d_stats from tia_hover_instance_stats; d_mark = n*h*w bytes of scratch;
 *          d_meta [n, max_inst+1, 4] i32 = start x, start y, vertex count, offset of the first vertex
 *          (exclusive prefix sum over the whole table); d_total[0] = number of vertices of all instances.
 *   write: d_points [capacity, 2] i32 (x, y) in image coordinates; instances whose vertices would not
 *          fit in `capacity` are left unwritten (pass capacity = *d_total).
 * The reference drops instances with fewer than 3 vertices (hovernet.py:695-699); that is the caller's job.
 */
int tia_hover_contour_scan(const int32_t* d_inst, int64_t n, int64_t h, int64_t w, int32_t max_inst,
                           const int64_t* d_stats, int8_t* d_mark, int32_t* d_meta, int64_t* d_total,
                           void* stream);
int tia_hover_contour_write(const int32_t* d_inst, int64_t n, int64_t h, int64_t w, int32_t max_inst,
                            const int64_t* d_stats, const int32_t* d_meta, int64_t capacity,
                            int32_t* d_points, void* stream);


/* =======================================================================================
 * Semantic-segmentation stitching (models/engine/semantic_segmentor.py:1141-1263,1398-1534)
 * ===================================================================================== */

/*
 * Horizontal merge of one patch row (merge_batch_to_canvas / merge_horizontal): every pixel of the
 * row canvas sums, in patch order, the blocks covering it; all-zero blocks are skipped
 * (:1178-1179); x-extents are clipped to the canvas width.
 *   d_blocks [n,oh,ow,c] f32   d_xs [n] i32 (output x0 of each block, ascending)
 *   d_row [oh,W,c] f32 (overwritten)   d_cnt [oh,W] u8 (overwritten)   d_flags [n] i32 scratch
 */
int tia_canvas_row_merge_f32(const float* d_blocks, const int32_t* d_xs, int64_t n, int64_t oh, int64_t ow,
                             int64_t c, int64_t width, float* d_row, uint8_t* d_cnt, int32_t* d_flags,
                             void* stream);

/*
 * Vertical merge + normalisation + argmax for canvas rows [y_begin, y_end)
 * (merge_vertical_chunkwise + post_process_wsi): value = rowA[y-ysA] (+ rowB[y-ysB] where the next
 * patch row overlaps), prob = value / max(count,1) (f32), prediction = argmax over channels (uint8).
 * d_row_b may be NULL.  d_probs ([H,W,c] f32) may be NULL.
 */
int tia_canvas_finalize_f32(const float* d_row_a, const uint8_t* d_cnt_a, int64_t ys_a, const float* d_row_b,
                            const uint8_t* d_cnt_b, int64_t ys_b, int64_t oh, int64_t width, int64_t c,
                            int64_t y_begin, int64_t y_end, float* d_probs, uint8_t* d_pred, void* stream);


/* =======================================================================================
 * Reinhard colour normalisation (tools/stainnorm.py:222-367): OpenCV 8-bit RGB<->Lab
 * ===================================================================================== */

/* Fixed-point tables of OpenCV's RGB2Lab_b / Lab2RGBinteger (built on the host once). */
typedef struct tia_lab_tables {
    uint16_t gamma[256];      /* sRGBGammaTab_b                                   */
    uint16_t cbrt[3072];      /* LabCbrtTab_b                                     */
    uint16_t lab_y[256];      /* LabToYF_b[2*i]                                   */
    uint16_t lab_ify[256];    /* LabToYF_b[2*i+1]                                 */
    uint8_t inv_gamma[4096];  /* sRGBInvGammaTab_b                                */
    int32_t c_fwd[9];         /* 12-bit sRGB->XYZ/whitepoint                      */
    int32_t c_inv[9];         /* 12-bit XYZ*whitepoint->sRGB                      */
} tia_lab_tables;

/* Per-image histograms of the 8-bit Lab channels, ACCUMULATED into d_hist [n,3,256] u32
 * (cv2.cvtColor(RGB2LAB) + cv2.meanStdDev, stainnorm.py:309-315,362-364). */
int tia_lab_hist_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                    uint32_t* d_hist, void* stream);

/* out = LAB2RGB(lut[c][ RGB2LAB(img)[c] ]) with one 3x256 uint8 LUT per image: the whole
 * ReinhardNormalizer.transform per-pixel chain (stainnorm.py:272-340) folded into tables. */
int tia_reinhard_apply_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                          const uint8_t* d_lut /* [n,3,256] */, uint8_t* d_out, void* stream);

/* Per-image Lab statistics and the 3x256 LUTs of tia_reinhard_apply_u8 from the histograms of
 * tia_lab_hist_u8, entirely on the device (stainnorm.py:277-292,336-339,362-367):
 * mean / population std of the float32 channels (cv2.meanStdDev) in f64, then the reference's float32
 * chain per Lab byte.  d_chan_vals [3,256] f32 = channel value of every Lab byte (L/2.55, a-128, b-128);
 * target_means / target_stds: host pointers to 3 doubles each (copied at launch).
 * d_meanstd [n,6] f64 (means then stds) and d_flags [n] i32 (bit 0: a zero std, where the reference raises
 * ZeroDivisionError) are optional outputs; d_flags must be zero-initialised. */
int tia_reinhard_luts(const uint32_t* d_hist, int64_t n, const float* d_chan_vals, const double* target_means,
                      const double* target_stds, uint8_t* d_lut, double* d_meanstd, int32_t* d_flags, void* stream);

/* cv2.cvtColor 8-bit RGB2LAB (dir 0) / LAB2RGB (dir 1) of npix pixels. */
int tia_lab_convert_u8(const uint8_t* d_src, int64_t npix, const tia_lab_tables* d_tables, int32_t dir,
                       uint8_t* d_dst, void* stream);

/* ReinhardNormalizer.transform in ONE launch for a batch of patches (stainnorm.py:272-367): per patch RGB->Lab, the channel
 * moments (cv2.meanStdDev, float64 in NumPy's pairwise order), the three float32 byte tables, table -> Lab->RGB.  The Lab image is
 * parked as one dword per pixel in a per-workgroup slot of d_workspace (tia_reinhard_workspace_bytes; it stays in L2 / Infinity
 * Cache), so HBM carries 3 bytes in + 3 bytes out per pixel.  Shapes it does not take (h*w not a multiple of 1024 or above 2^18
 * pixels, buffers not 16-byte aligned) return TIA_ESIZE: use tia_lab_hist_u8 + tia_reinhard_luts + tia_reinhard_apply_u8, which
 * split a large image over many workgroups.  d_chan_vals: float32 [3,256] channel value of every Lab byte (host-built, :295-315);
 * target_means / target_stds: host double[3]; d_meanstd (nullable) [n,6]; d_flags (nullable) [n]: bit 0 = zero std (the reference
 * raises ZeroDivisionError). */
size_t tia_reinhard_workspace_bytes(int64_t n, int64_t h, int64_t w);
int tia_reinhard_transform_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                              const float* d_chan_vals, const double* target_means, const double* target_stds, uint8_t* d_out,
                              double* d_meanstd, int32_t* d_flags, void* d_workspace, size_t workspace_bytes, void* stream);

/* ReinhardNormalizer.get_mean_std of n images in one launch (stainnorm.py:343-367): d_meanstd [n,6] = means of L/2.55, a-128, b-128,
 * then the population standard deviations.  Same shape limits as tia_reinhard_transform_u8 (TIA_ESIZE otherwise). */
int tia_lab_moments_u8(const uint8_t* d_img, int64_t n, int64_t h, int64_t w, const tia_lab_tables* d_tables,
                       const float* d_chan_vals, double* d_meanstd, int32_t* d_flags, void* stream);


/* out[i, j] = lut[i, img[i, j]]: one 256-byte table per image -- the intensity map of contrast_enhancer
 * (utils/misc.py:436-444: rescale_intensity over the [p2, p98] window, truncated to uint8) applied to a batch.
 *   d_img, d_out [n, len] u8   d_lut [n, 256] u8   n <= 65535 */
int tia_lut_apply_u8(const uint8_t* d_img, int64_t n, int64_t len, const uint8_t* d_lut, uint8_t* d_out, void* stream);

/* Box down-sampling of an HWC uint8 image by an integer factor: integer box sum * (1.0f / area), rounded half to even --
 * cv2.INTER_AREA at an integer scale, which is what the reference's slide thumbnail goes through
 * (utils/transforms.py imresize; wsicore/wsireader.py:1735-1786 tissue_mask -> slide_thumbnail).
 *   d_src [h,w,c]   d_out [h/factor, w/factor, c] */
int tia_box_downsample_u8(const uint8_t* d_src, int64_t h, int64_t w, int64_t c, int64_t factor, uint8_t* d_out, void* stream);

/* =======================================================================================
 * CNN epilogues (NHWC activations).  The convolutions themselves run in MIOpen; with BatchNorm
 * folded into the weights every conv is followed by bias (+ residual) + ReLU, which PyTorch
 * executes as 2-3 separate full-tensor passes (models/architecture/vanilla.py:300-316 forward).
 * ===================================================================================== */
#define TIA_DT_F32 0
#define TIA_DT_F16 1
#define TIA_DT_BF16 2

/* In place: x[r,c] = act(x[r,c] + bias[c] (+ residual[r,c])), one pass.  c % 8 == 0. */
int tia_bias_act_nhwc(void* d_x, const void* d_bias, const void* d_residual, int64_t rows, int64_t c,
                      int32_t dtype, int32_t relu, void* stream);

/* out = maxpool3x3/s2/p1(relu(x + bias)) for the ResNet stem, one pass.
 * x [n,h,w,c] -> out [n,(h+1)/2,(w+1)/2,c].  c % 8 == 0. */
int tia_bias_relu_maxpool_nhwc(const void* d_x, const void* d_bias, int64_t n, int64_t h, int64_t w,
                               int64_t c, int32_t dtype, void* d_out, void* stream);

/* Patch reads from an in-memory slide level (WSIPatchDataset.__getitem__, models/dataset/dataset_abc.py:418-448;
 * tools/patchextraction.py:356-461 supplies the bounds): out[i] = slide[y0:y0+ph, x0:x0+pw] for d_bounds[i] =
 * (x0, y0, x1, y1) in baseline pixels, with `pad` (255) wherever the region leaves the slide.
 *   d_slide [sh,sw,c] u8   d_bounds [m,4] i32   d_out [m,ph,pw,c] u8 (ph*pw*c % 4 == 0, m <= 65535) */
int tia_gather_patches_u8(const uint8_t* d_slide, int64_t sh, int64_t sw, int64_t c, const int32_t* d_bounds,
                          int64_t m, int64_t ph, int64_t pw, int32_t pad, uint8_t* d_out, void* stream);

/* =======================================================================================
 * ResNet convolutions on the matrix cores (models/architecture/vanilla.py:300-316 -> torchvision BasicBlock)
 * ===================================================================================== */

/* OIHW float32 weights -> the implicit GEMM's B matrix [kh][kw][cin][cout] (done once per model). */
int tia_conv_pack_weights_f32(const float* d_w_oihw, int64_t cout, int64_t cin, int64_t kh, int64_t kw,
                              float* d_packed, void* stream);

/* y = act(conv2d(x, w) + bias [+ residual]) for NHWC float32 tensors: implicit GEMM on v_mfma_f32_32x32x2_f32 (f32 in,
 * f32 accumulate: the reference's float32 arithmetic, vanilla.py:242), BatchNorm already folded into w / bias.
 *   d_x [n,h,w,cin]   d_w_packed [kh,kw,cin,cout]   d_bias [cout] or NULL   d_residual [n,ho,wo,cout] or NULL
 *   d_y [n,ho,wo,cout], ho = (h + 2 pad - kh) / stride + 1.   cin % 32 == 0, cout % 64 == 0, 16-byte aligned x / w. */
int tia_conv2d_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                        float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                        int64_t kw, int64_t stride, int64_t pad, int32_t relu, void* stream);

/* The same convolution with the zero border given explicitly: `pad_top` / `pad_left` rows / columns of zeros in front,
 * and as many behind as the requested output size [ho, wo] reaches (taps outside the image read as zeros).  This is how
 * TensorFlow-style "same" padding of a strided convolution (0 in front, 1 behind: HoVer-Net's TFSamepaddingLayer,
 * models/architecture/hovernet.py:30-69) and "valid" convolutions (pad 0, ho = (h - kh) / stride + 1) are expressed.
 * Requires pad_top < kh, pad_left < kw and (ho - 1) * stride - pad_top < h (likewise for columns). */
int tia_conv2d_nhwc_f32_ex(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                           float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                           int64_t kw, int64_t stride, int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo,
                           int32_t relu, void* stream);

/* Which block geometry tia_conv2d_nhwc_f32(_ex) / tia_conv2d_nhwc_h use for a 3x3 / stride-1 convolution of an [h, w] map to
 * [ho, wo] in float32 (diagnostics for tests and profiles; host only, no launch).  Returns 0: the slice implicit-GEMM kernel;
 * 1: tap reuse on 16 x 16 pixel blocks; 2: tap reuse, two images of at most 8 x 8 per block; 3: tap reuse on bands of geom[1]
 * rows of a geom[0]-column strip of the batch stacked into one tall image with a zero row between neighbours; 4: the same with
 * bands of geom[1] REAL rows (the zero rows are in the block's LDS patch, not among its GEMM rows).
 * geom[0..3] = strip width, rows per band, LDS row pitch (16-byte units), strips per image row (zeros unless 3 / 4). */
int tia_conv3x3_geometry(int64_t h, int64_t w, int64_t ho, int64_t wo, int64_t pad_top, int64_t pad_left, int32_t geom[4]);

/* Convolution over a THIN input (c * kw <= 32, e.g. the 3-channel 7x7 stem of HoVer-Net, models/architecture/hovernet.py:
 * 287-300 `conv0`): d_x [n,h,w,c] float32 NHWC, ALREADY padded horizontally by the caller ((wo-1)*stride + kw <= w, and at least 32 floats from the last output
 * column's first tap to the end of its row: (w - (wo-1)*stride) * c >= 32); rows are
 * padded by pad_top / ho like tia_conv2d_nhwc_f32_ex.  d_w_packed: [kh][32][cout] float32 with row kx * c + ch = w[o][ch][ky][kx]
 * and zero rows from kw * c on.  Same kernel and arithmetic as tia_conv2d_nhwc_f32 (K = 32 * kh). */
int tia_conv2d_thin_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, float* d_y, int64_t n, int64_t h,
                             int64_t w, int64_t c, int64_t cout, int64_t kh, int64_t kw, int64_t stride, int64_t pad_top,
                             int64_t ho, int64_t wo, int32_t relu, void* stream);

/* 1x1 convolution with FEW output channels (the class heads: cin = 64 -> cout <= 8; hovernet.py:196-199 `u0/conv`,
 * unet.py:336 `clf`), optionally with the BatchNorm + ReLU that precedes it applied on load:
 *   y[p][o] = bias[o] + sum_c w[o][c] * pre(x[p][c]),  pre(v) = relu(v * pre_scale[c] + pre_shift[c]) or v (both NULL).
 * d_x [npix,64], d_w [cout][64] (the OIHW tensor), d_y [npix,cout], all float32; HBM-bound (one read of x). */
int tia_conv1x1_head_nhwc_f32(const float* d_x, int64_t npix, const float* d_w, const float* d_bias, const float* d_pre_scale,
                              const float* d_pre_shift, int32_t cout, float* d_y, void* stream);

/* tia_conv2d_nhwc_f32_ex with a second, post-activated output produced in the same epilogue:
 *   v  = act(conv(x, w) + bias [+ residual])      -> d_y   (may be NULL when only d_y2 is wanted)
 *   y2 = relu(v * post_scale[c] + post_shift[c])   -> d_y2  (required)
 * i.e. the BatchNorm + ReLU that follows a residual sum in a pre-activation network (HoVer-Net's next-unit "preact" /
 * "blk_bna", models/architecture/hovernet.py:100-147), applied while the sum is still in registers. */
int tia_conv2d_post_nhwc_f32(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                             float* d_y, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh,
                             int64_t kw, int64_t stride, int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo,
                             int32_t relu, const float* d_post_scale, const float* d_post_shift, float* d_y2,
                             void* stream);

/* Winograd F(2x2, 3x3) form of the 3x3 / stride-1 float32 convolution (conv3x3_wino.hip; reference call sites: the 3x3 layers
 * behind CNNModel.forward, models/architecture/vanilla.py:242-253,300-316): 16 instead of 36 multiplies per 2 x 2 outputs on the same
 * float32 MFMA instruction, input transform in registers, weights transformed once at pack time in float64.  Same contract as
 * tia_conv2d_nhwc_f32_ex for kh = kw = 3, stride = 1 (pad_top / pad_left in 0..2, ho / wo given; bias + residual + ReLU fused), but
 * NOT bit-identical to it: float32 Winograd rounds differently from a direct float32 convolution (<= ~1e-5 relative, asserted in
 * tests/test_engine.py) -- which is why nothing dispatches to it implicitly.  cin % 16 == 0, cout % 64 == 0.
 *   tia_conv_pack_weights_wino_f32: OIHW [cout][cin][3][3] -> U = G g G^T, 16 * cin * cout floats in the kernel's stage layout
 *   ([pos 16][cin/16][h8 2][cout/64][hi 2][64 cout][4 channels], channel = 16 cs + 8 h8 + 4 hi + c4). */
int tia_conv_pack_weights_wino_f32(const float* d_w_oihw, int64_t cout, int64_t cin, float* d_packed, void* stream);
int tia_conv3x3_wino_nhwc_f32(const float* d_x, const float* d_u_packed, const float* d_bias, const float* d_residual, float* d_y,
                              int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t pad_top, int64_t pad_left,
                              int64_t ho, int64_t wo, int32_t relu, void* stream);

/* Host-only query (no launch): which kernel tia_conv2d_nhwc_f32[_ex] runs a float32 convolution of this shape on --
 * 0: conv_mfma_f32_kernel (register-staged 128-pixel slices), 1: conv3x3_spatial_kernel (tap reuse; tia_conv3x3_geometry says
 * which block geometry), 2: conv1x1_ring_kernel (LDS-DMA ring over 256-pixel blocks: 1x1, and kh x kw taps gathered).  For
 * tests and bench.py (they ask instead of mirroring the dispatch rule).  It runs the entry point's own shape checks, batch split
 * (< 2 GiB of input per launch) and decision function: shapes the entry point rejects return the same negative code (TIA_EINVAL /
 * TIA_ESIZE, e.g. cin % 32 != 0); a batch that is split goes in EQUAL groups (ceil(n / k) images for the smallest k that fits:
 * 4096 images of 1 MiB run as 1366 + 1366 + 1364) and the answer is the route of that group size.  The ring's and the bands' fill
 * rules use the CU count of the calling thread's current device (256 without a device). */
int tia_conv2d_route_f32(int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                         int64_t pad_top, int64_t pad_left, int64_t ho, int64_t wo);

/* 1x1 convolution (any stride, no padding) whose INPUT is activated on load:
 *   y = act(conv1x1(relu(x * pre_scale[c] + pre_shift[c]), w) + bias [+ residual])
 * -- the "preact/bn" + ReLU in front of conv1 of HoVer-Net's residual units 2..n (models/architecture/hovernet.py:100-147),
 * applied between the global load and the LDS store of the GEMM's A operand (product and sum rounded separately, like
 * batch_norm + relu): the unit reads the raw residual sum of the previous unit and the activated copy is never written.
 * pre_scale / pre_shift [cin], 16-byte aligned; other arguments as tia_conv2d_nhwc_f32_ex (ho = ceil(h / stride)). */
int tia_conv1x1_pre_nhwc_f32(const float* d_x, const float* d_pre_scale, const float* d_pre_shift, const float* d_w_packed,
                             const float* d_bias, const float* d_residual, float* d_y, int64_t n, int64_t h, int64_t w,
                             int64_t cin, int64_t cout, int64_t stride, int32_t relu, void* stream);

/* The ResNet stem in one kernel: y = maxpool3x3/s2/p1(relu(conv7x7/s2/p3(X) + bias)), 3 -> 64 channels, BatchNorm folded
 * into w / bias (CNNModel.forward -> torchvision conv1 / bn1 / relu / maxpool, models/architecture/vanilla.py:300-316).
 *   x_is_u8 != 0: d_x is the uint8 NHWC patch batch and X = x / 255 in float32 (correctly rounded division) -- `ToTensor`
 *   (models/dataset/classification.py:27-32) and the float32 cast of `_infer_batch` (vanilla.py:242-245) happen on load;
 *   x_is_u8 == 0: d_x is float32 NHWC and X = x.
 *   d_x [n,h,w,3]   d_w_packed [148,64] from tia_stem_pack_weights_f32   d_bias [64]
 *   d_y [n,hp,wp,64] of y_dtype (TIA_DT_F32; TIA_DT_F16 / TIA_DT_BF16: rounded once, for the half-precision trunk),
 *   hp = ((h-1)/2)/2 + 1 (likewise wp); arithmetic: a float32 fmaf chain in (ky, kx, c) order on the matrix cores.
 *   d_conv_out (may be NULL): also write relu(conv + bias) BEFORE the pooling, [n,ho,wo,64] float32, ho = (h-1)/2 + 1 -- the
 *   first skip connection of the UNet decoder (models/architecture/unet.py:356-372, ResNetEncoder features). */
int tia_stem_conv7x7_pool_nhwc(const void* d_x, int32_t x_is_u8, const float* d_w_packed, const float* d_bias, void* d_y,
                               int32_t y_dtype, float* d_conv_out, int64_t n, int64_t h, int64_t w, void* stream);

/* The same stem on the HALF matrix cores, for compute_dtype = float16 | bfloat16 of the engines (an extension: the reference runs
 * float32; `model.half()(ToTensor(x).half())` is what it mirrors): x / 255 and the weights rounded to `dtype`, float32
 * accumulation (v_mfma_f32_32x32x16_f16 / _bf16, K = 7 * 24 = 168 padded to 176), bias + ReLU + max-pool in float32, one rounding
 * of the pooled result.  d_w_packed_h: [22][64][8] halves from tia_stem_pack_weights_h (row k = 24 ky + 3 kx + c of the
 * OIHW float32 tensor); d_y [n,hp,wp,64] of `dtype` (TIA_DT_F16 | TIA_DT_BF16). */
int tia_stem_pack_weights_h(const float* d_w_oihw, int32_t dtype, void* d_packed, void* stream);
int tia_stem_conv7x7_pool_nhwc_h(const void* d_x, int32_t x_is_u8, const void* d_w_packed_h, const float* d_bias, void* d_y,
                                 int32_t dtype, int64_t n, int64_t h, int64_t w, void* stream);
/* OIHW [64,3,7,7] float32 -> [148,64]: rows (ky, kx, c), one zero row at the end. */
int tia_stem_pack_weights_f32(const float* d_w_oihw, float* d_packed, void* stream);

/* Half-precision sibling of tia_conv2d_nhwc_f32 for `compute_dtype="float16"|"bfloat16"` runs of the engines (an extension:
 * the reference computes in float32, vanilla.py:242; results are held to its own 1e-3 tolerance on the probabilities,
 * tests/engines/test_patch_predictor.py:712-722): implicit GEMM on v_mfma_f32_32x32x16_f16 / _bf16, float32 accumulate;
 * bias + residual + ReLU in float32, ONE rounding to half on the way out.
 *   d_x [n,h,w,cin] half   d_w_packed from tia_conv_pack_weights_h   d_bias [cout] float32 or NULL
 *   d_residual [n,ho,wo,cout] half or NULL   d_y [n,ho,wo,cout] half.   cin % 32 == 0, cout % 64 == 0, 16-byte aligned. */
int tia_conv2d_nhwc_h(const void* d_x, const void* d_w_packed, const float* d_bias, const void* d_residual, void* d_y,
                      int64_t n, int64_t h, int64_t w, int64_t cin, int64_t cout, int64_t kh, int64_t kw, int64_t stride,
                      int64_t pad, int32_t dtype, int32_t relu, void* stream);
/* OIHW float32 weights -> [kh][kw][cin/8][cout][8] halves of `dtype` (a lane's 8 k-values of its column contiguous). */
int tia_conv_pack_weights_h(const float* d_w_oihw, int64_t cout, int64_t cin, int64_t kh, int64_t kw, int32_t dtype,
                            void* d_packed, void* stream);

/* y = act(x * scale[c] + shift[c]) on NHWC float32 ([rows, c], c % 4 == 0; y may alias x): an inference-mode
 * BatchNorm (+ ReLU) that sits in FRONT of a convolution and therefore cannot be folded into one -- the pre-activation
 * units of HoVer-Net (models/architecture/hovernet.py:72-261: "preact_bna" / "blk_bna"). */
int tia_scale_shift_act_nhwc_f32(const float* d_x, const float* d_scale, const float* d_shift, float* d_y, int64_t rows,
                                 int64_t c, int32_t relu, void* stream);

/* The same on a view: d_x points at element [0,0,0,0] of a window of a wider NHWC buffer (a channel prefix and a spatial
 * crop), strides in elements (multiples of 4, 16-byte aligned base, x_pixel_stride >= c); y [n,h,w,c] dense.  Lets the
 * dense units of HoVer-Net (hovernet.py:72-98) grow their feature stack in place instead of re-concatenating it. */
int tia_scale_shift_act_view_nhwc_f32(const float* d_x, int64_t x_image_stride, int64_t x_row_stride, int64_t x_pixel_stride,
                                      const float* d_scale, const float* d_shift, float* d_y, int64_t n, int64_t h,
                                      int64_t w, int64_t c, int32_t relu, void* stream);

/* Grouped "valid" k x k convolution, stride 1, 32 input and 8 output channels per group (the second convolution of
 * HoVer-Net's dense units, hovernet.py:86-88: Conv2d(128, 32, k, groups=4)); float32 NHWC.
 *   d_x [n,h,w,groups*32] dense;  d_w_packed [groups][k][k][32][8] (from OIHW [groups*8, 32, k, k]);
 *   y: element [b, oy, ox, c] at d_y + b*y_image_stride + oy*y_row_stride + ox*y_pixel_stride + c (strides in
 *   elements, multiples of 4), ho = h - k + 1 -- so the result can land in a channel slice of a wider buffer.
 * Other channel counts -> TIA_ESIZE. */
int tia_grouped_conv_valid_nhwc_f32(const float* d_x, const float* d_w_packed, float* d_y, int64_t y_image_stride,
                                    int64_t y_row_stride, int64_t y_pixel_stride, int64_t n, int64_t h, int64_t w,
                                    int64_t groups, int64_t cin_per_group, int64_t cout_per_group, int64_t k, void* stream);

/* out[b, Y, X, :] = x[b, Y/2, X/2, :] + y[b, Y, X, :] on NHWC float32: nearest x2 upsampling fused with the decoder's
 * skip-connection add (models/architecture/hovernet.py:447-449, utils.py:202-243).  x [n,h,w,c]; y a (possibly
 * centre-cropped) view of an NHWC tensor with contiguous channels: d_y points at its first element,
 * y_image_stride / y_row_stride in elements (multiples of 4, 16-byte aligned base); out [n,2h,2w,c]; c % 4 == 0. */
int tia_upsample2x_add_nhwc_f32(const float* d_x, const float* d_y, int64_t y_image_stride, int64_t y_row_stride,
                                float* d_out, int64_t n, int64_t h, int64_t w, int64_t c, void* stream);
/* The same followed by relu(. * scale[c] + shift[c]) (both NULL: plain sum): up-sampling, skip add and the pre-activation of
 * a UNet decoder block (models/architecture/unet.py:193-240, 356-417) in one pass. */
int tia_upsample2x_add_act_nhwc_f32(const float* d_x, const float* d_y, int64_t y_image_stride, int64_t y_row_stride,
                                    const float* d_scale, const float* d_shift, float* d_out, int64_t n, int64_t h,
                                    int64_t w, int64_t c, void* stream);

/* =======================================================================================
 * All borders of binary planes: cv2.findContours(layer, RETR_TREE, CHAIN_APPROX_NONE | _SIMPLE)
 * (models/architecture/hovernetplus.py:222-226, HoVerNetPlus._get_layer_info)
 * ===================================================================================== */

/* Raster-first pixel (flat index, 0x7f7f7f7f if absent) and frame contact of every label 1..kmax of every plane:
 * with tia_ccl_label_i32 (8-connected foreground, 4-connected background) this yields the start pixel of every outer
 * border (a component's first pixel) and of every hole border (left of an enclosed background component's first pixel).
 *   d_labels [n,h,w] i32   d_first, d_edge [n, kmax+1] i32 */
int tia_label_first_pixel_i32(const int32_t* d_labels, int64_t n, int64_t h, int64_t w, int32_t kmax,
                              int32_t* d_first, int32_t* d_edge, void* stream);

/* Border following (Suzuki & Abe 1985, step 3 = OpenCV icvFetchContour), one lane per border.
 *   d_starts [nb,4] i32 = plane, x0, y0, is_hole.  d_points == NULL: count pass, d_counts[nb] = points per border.
 *   Otherwise border e writes its (x, y) pairs at d_points + 2*d_offsets[e] (capacity = total pairs allocated).
 *   simple != 0: CHAIN_APPROX_SIMPLE (direction changes only); 0: CHAIN_APPROX_NONE (every border pixel). */
int tia_border_trace_u8(const uint8_t* d_mask, int64_t n, int64_t h, int64_t w, const int32_t* d_starts, int64_t nb,
                        int32_t simple, int32_t* d_counts, const int64_t* d_offsets, int64_t capacity,
                        int32_t* d_points, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TIATOOLBOX_AMD_H */
