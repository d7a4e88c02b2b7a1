/* libbsr_hip — C ABI of the MI355X-native GSC generator forward pass.
 *
 * The reference has no FFI / plugin boundary: the hot path is entered through a Python call on a
 * tf.keras.Model,
 *     deshadow_img_gs, deshadow_img_c, _, mask_pred = self.gen(img, uv, reg, chuck=1, training=False)
 * (/root/reference/train_test_GSC.py:422, :871; Generator.call at /root/reference/model.py:228-290).
 * This header is what a binding for that call site binds instead (INTEGRATION.md shows the ctypes stub):
 * plain pointers and sizes, no framework types.  All tensor pointers are DEVICE pointers owned by the
 * caller (NHWC float32, dense); the library owns only its packed weights and its activation workspace.
 *
 * Threading: a handle is bound to one device and is not re-entrant; bsr_forward is asynchronous on the
 * given hipStream_t and the caller synchronises.  Multi-GPU = one handle per device (deployment: one process per GPU;
 * one process may also hold handles on several devices — per-device launch state is kept per device ordinal).  Every
 * entry point switches to the handle's device and restores the caller's current device before returning.
 * Every function returns 0 on success or a non-zero code; bsr_last_error() describes the last failure
 * of the calling thread.
 */
#ifndef BSR_HIP_H_
#define BSR_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsr_handle bsr_handle;

#define BSR_OK 0
#define BSR_ERR_ARG 1      /* bad argument / shape */
#define BSR_ERR_BLOB 2     /* malformed or mismatching packed-weight blob */
#define BSR_ERR_HIP 3      /* a HIP runtime call failed */
#define BSR_ERR_STATE 4    /* probe requested before any forward, unknown probe name, ... */
#define BSR_ERR_RANGE 5    /* 16-bit modes: an activation did not fit fp16 (|x| >= 65520) — see bsr_check_range */

#define BSR_DTYPE_F32 0
#define BSR_DTYPE_F16 1     /* BASELINE configs[3]: fp16 operands (fp32 accumulate) on the 3x3-conv path via v_mfma_f32_32x32x16_f16, with the tensors between
                               those layers kept as fp16 in the library's workspace; trunk, 1x1 / attention kernels (split precision) and all
                               inputs / outputs of the ABI stay fp32.  ~1.3e-3 absolute accuracy */
#define BSR_DTYPE_F32X3 2   /* split-precision fp32: every operand of the matrix kernels is split into hi + lo fp16 halves at staging time and contracted
                               with three fp16 matrix instructions (hi.hi + hi.lo + lo.hi, fp32 accumulate) at 16/3 of the fp32 matrix rate.
                               Accuracy: x = hi + lo to 2^-22 |x| while |x| >= 2^-3; smaller operands carry an absolute error of up to 2^-25
                               (lo is an unscaled fp16 subnormal), i.e. 2^-17..2^-20 relative for typical folded weights — the 1e-3 end-to-end
                               parity bar holds with the fp32 path's margin (measured 4.5e-6), it is NOT 2^-22 per product.  Activations / outputs
                               stay fp32.  Range: |activation| must stay below 65504 (fp16 max); a violation is DETECTED on the device and
                               reported as BSR_ERR_RANGE (bsr_check_range), never silently turned into inf / NaN outputs. */

/* Replaces Generator() construction + tf.train.Checkpoint(generator=...).restore(...)
 * (/root/reference/train_test_GSC.py:120, :143-148, :365, :845).
 * packed_weights: HOST pointer to the blob written by blindshadowremoval_amd.pack.pack_generator()
 * (BatchNorm folded, MFMA-friendly layout); it is copied to the device, the caller may free it.
 * dtype: BSR_DTYPE_F32 (the arithmetic type of the measured path), BSR_DTYPE_F16 or BSR_DTYPE_F32X3; the blob must have been packed
 * for the same dtype (pack_generator(weights, dtype): the 16-bit modes carry fp16 weight planes for the 3x3-conv layers). */
int bsr_create(bsr_handle** out, int device, const void* packed_weights, size_t nbytes, int dtype);

/* Replaces Generator.call(inputs, uv, reg, chuck, training=False) (/root/reference/model.py:228-290).
 * inputs, uv : [B,H,W,3] float32 device pointers (reg / chuck are unused by the reference's GSC forward).
 * gs [B,H,W,1], con_rgb [B,H,W,3], mask22 [B,H,W,3], dif [B,H,W,1] : caller-allocated outputs, in the order
 * of the reference's return statement (model.py:290).  H % 32 == 0 and W % 256 == 0 (reference: 256 x 256).
 * stream: a hipStream_t (NULL = default stream). */
int bsr_forward(bsr_handle* h, const float* inputs, const float* uv, int B, int H, int W,
                float* gs, float* con_rgb, float* mask22, float* dif, void* stream);

/* bsr_forward with the two outputs the reference's callers consume (train_test_GSC.py:871-873: deshadow_img_c, mask_pred) written as
 * ONE tensor con_rgb_dif [B,H,W,4] = con_rgb(3) | dif(1): the payload of the multi-GPU output all-gather (one 16-byte store per pixel,
 * no separate packing pass).  Values are bit-identical to bsr_forward's con_rgb / dif. */
int bsr_forward_packed(bsr_handle* h, const float* inputs, const float* uv, int B, int H, int W,
                       float* gs, float* con_rgb_dif, float* mask22, void* stream);

/* TSM variant (BASELINE config 5): replaces Generator.call(inputs, uv, reg, frame, share, chuck, training=False) of
 * /root/reference/model_with_TSM.py:261-325 (call site /root/reference/train_with_TSM.py:676).  The handle must have been
 * created from TSM weights (res_stack/0/conv1 with 291 input channels).  reg: [B,H,W,6] = reg_in(3) | reg_out(3) offset
 * fields (image-fraction units); consecutive groups of `frame` images share features through the UV/offset warp
 * (ShareLayer, model_with_TSM.py:199-229; warp.py:134-165).  share = 0 reproduces the tf.cond(share, ...) false branch. */
int bsr_forward_tsm(bsr_handle* h, const float* inputs, const float* uv, const float* reg, int B, int H, int W, int frame, int share,
                    float* gs, float* con_rgb, float* mask22, float* dif, void* stream);

/* Range guard of BSR_DTYPE_F32X3 / BSR_DTYPE_F16.  Those modes convert fp32 activations to fp16 operands inside the kernels; a value of
 * magnitude >= 65520 would become inf (and its lo half NaN) where the fp32 path stays finite.  Every converting kernel checks what
 * it converts and sets a sticky, host-visible flag on the handle.  bsr_check_range synchronises `stream` (the stream the forwards
 * ran on), returns BSR_ERR_RANGE if any forward since the last call overflowed — the outputs of those forwards must be discarded
 * and the inputs re-run on a BSR_DTYPE_F32 handle — and clears the flag.  Without a call the condition is still not silent:
 * bsr_forward / bsr_forward_tsm refuse with BSR_ERR_RANGE once a completed forward has raised the flag.  Always BSR_OK on a
 * BSR_DTYPE_F32 handle.  (NaN inputs are not a range error; they propagate to the outputs.) */
int bsr_check_range(bsr_handle* h, void* stream);
/* The same condition WITHOUT synchronising and WITHOUT clearing it: BSR_ERR_RANGE if any forward that has completed so far raised the
 * flag.  For pipelined callers that keep several forwards in flight and wait on their own events (FSRNet's loops): after the event of
 * forward k, a set flag means forward k — or one submitted after it that has already finished — overflowed.  ABI 5. */
int bsr_peek_range(bsr_handle* h);

/* Bytes of activation workspace the library holds for a batch of B HxW images (grown lazily by
 * bsr_forward; growth synchronises the stream — call bsr_reserve first to keep forwards allocation-free). */
size_t bsr_workspace_bytes(int B, int H, int W);                              /* GSC channel plan */
size_t bsr_handle_workspace_bytes(const bsr_handle* h, int B, int H, int W);  /* the plan of THIS handle's variant (GSC or the wider TSM one) */
int bsr_reserve(bsr_handle* h, int B, int H, int W);

/* Test hook: copy a named intermediate of the LAST forward (dense NHWC, real channel count) into dst
 * (device pointer, capacity cap_floats).  shape4 receives [B,H,W,C].  Names: x1 x2 x3 x0 res0..res5 up1 up2
 * y d32 bmask xh f1 f2 f y3x<i> att<i> — att<i> (the attention output of block i) only exists when the last forward ran attention and the
 * `w` GEMM as separate launches (small batches, the 16-bit modes, BSR_FUSE_ATTW=0); after a fused forward it never left LDS and the
 * probe returns BSR_ERR_STATE.  shape4 is filled even when cap_floats is too small (BSR_ERR_ARG), so a caller can
 * size its buffer with a first call of capacity 0. */
int bsr_probe(bsr_handle* h, const char* name, float* dst, size_t cap_floats, int shape4[4], void* stream);

/* Per-kernel-class device time (ms) of the last forward run with timing enabled; classes are indexed
 * 0: conv3x3 (+stride-2), 1: transposed 3x3 (all but class 6), 2: conv1x1, 3: attention, 4: conv7 (stem + heads), 5: glue,
 * 6: the dominant kernel instantiation igemm_conv_kernel<3,3,1,true,4,32,4,1,1,2,32,1> (up2, up3, clr_up3).
 * bsr_set_timing(h, 1) makes every following forward record HIP events around each launch on the
 * forward's stream (adds host overhead; off by default). */
#define BSR_NUM_CLASSES 7
int bsr_set_timing(bsr_handle* h, int enable);
int bsr_get_timing(bsr_handle* h, float ms_per_class[BSR_NUM_CLASSES], int launches_per_class[BSR_NUM_CLASSES]);
/* The same events, launch by launch in issue order: bsr_timing_launches() entries; entry i = the layer name the launch computes
 * ("conv1", "down1", "res3.conv2", "res3.c3q" (conv3 + theta|phi|g), "res3.attention", "res3.w", "up2", "heads", "clr_conv1", glue
 * kernel names), its device time and its class.  bench.py derives the per-kernel roofline from these. */
int bsr_timing_launches(bsr_handle* h);
int bsr_timing_entry(bsr_handle* h, int i, char* name, size_t name_cap, float* ms, int* cls);

/* Input preparation for a batch of rows on the device: the per-sample work of the reference's test loaders
 * (/root/reference/dataset.py:619-638, 148-170; utils.py:356-433 face_crop_and_resize, :255-276 generate_face_region;
 * warp.py:194-232 generate_offset_map / generate_uv_map) after PNG decoding and Delaunay triangulation, which stay on the host.
 * d_blob: ONE device buffer holding, at 8-byte aligned offsets, B row records (csrc/prep_kernels.h PrepRow: image / ground-truth
 * RGB8 offsets and size, crop box, four triangle tables), `S` float64 grid coordinates (numpy.linspace(0, 1, S)) and the data
 * they point to — blindshadowremoval_amd/prep.py builds it.  out: [B,S,S,16] float32 = img3 | gt3 | uvm3 | reg_in3 | reg_out3 |
 * face1 (the packed layout FSRNet.test_step / test_step_FFHQ split, train_test_GSC.py:419,870); hull_tmp: [B,S,S] float32 scratch.
 * device: the GPU d_blob / out / hull_tmp live on (the call runs there whatever the caller's current device is); blob_bytes: size of
 * d_blob — the row and grid tables are checked against it (BSR_ERR_ARG), and the builder of the blob must keep every offset a row
 * record holds (img_off / gt_off + h*w*3, tri_off[k] + ntri[k]*144) inside it: prep.py does, before the upload.  ABI 4. */
int bsr_prep_rows(int device, const void* d_blob, size_t blob_bytes, size_t rows_off, size_t grid_off, int B, int S, float* out, float* hull_tmp,
                  void* stream);

/* PNG scanline reconstruction (RFC 2083 section 6) of n images on the device — what cv2.imread / PIL do after inflating a file
 * (/root/reference/dataset.py:151,622: the images parse_fn_test / parse_fn_test_FFHQ read), moved behind the copy to the device so that a
 * loader's worker stops at the inflated stream.  d_blob (device, blob_bytes): at items_off (8-byte aligned) n records
 * { int64 raw_off, out_off; int32 h, w, c, grey_out } — raw_off: h x (1 + w c) bytes of FILTERED scanlines (filter-type byte first; c = 1
 * grey, 3 RGB, 4 RGBA of an 8-bit non-interlaced file), out_off: where the RGB8 image [h][w][3] is written (grey replicated, alpha
 * dropped — PIL's convert("RGB"); grey_out != 0 with c = 1: one byte per pixel, [h][w] — the UCB masks); both inside the blob, validated by the caller like bsr_prep_rows' records: h <= 256 (one workgroup
 * per image, one thread per row, pixels on the anti-diagonal), w c >= 4, and 16 readable bytes of the blob in front of and behind every
 * filtered image (a thread reads its row four pixels at a time).  ABI 8. */
int bsr_png_unfilter(int device, void* d_blob, size_t blob_bytes, size_t items_off, int n, void* stream);

/* The output sink of the reference's loops on the device: replaces `cv2.imwrite(fname, strip)` of Logging.save_img
 * (/root/reference/utils.py:196-204; called per item from train_test_GSC.py:744-746 and :889-890) up to the write() itself.
 * pixels: [B,H,W,3] uint8 RGB strips (device).  out: B complete PNG FILE images, out_stride bytes apart (device or device-mapped
 * pinned memory), each exactly bsr_png_file_bytes(H, W) long: 8-bit truecolour, filter type 0, the zlib stream as stored deflate
 * blocks (lossless: any decoder returns `pixels`; size = raw size + 0.3 %), Adler-32 and chunk CRC-32s computed on the device.
 * scratch: bsr_png_scratch_bytes(B) bytes of device memory, 8-byte aligned (the per-file checksum accumulators and arrival tickets).
 * ABI 7: the scratch must be ZERO when it is first used (one hipMemset when it is allocated) and every call leaves it zero again — the
 * encoder is ONE launch whose last workgroup per file finishes the file and clears its accumulators (ABI 5-6 cleared them with a memset per
 * call and finished with a third launch).  A scratch that is not zero gives files with wrong checksums.  One call at a time per scratch.
 * W <= 5461; the Adler-32 sums are reduced per workgroup, so every admitted size (up to 65535 x 5461 pixels) checksums correctly.
 * Asynchronous on `stream`. */
size_t bsr_png_file_bytes(int H, int W);
size_t bsr_png_scratch_bytes(int B);
int bsr_png_encode(int device, const unsigned char* pixels, int B, int H, int W, unsigned char* out, size_t out_stride, void* scratch,
                   void* stream);

/* The same files straight from the FIGURES: replaces Logging.get_imgs + cv2.imwrite (/root/reference/utils.py:180-204: clip to [0,1],
 * x 255, to uint8, figures side by side) for a batch.  The strip is n_figs figures of H x Wf pixels side by side (W = n_figs * Wf);
 * figure k is float32 [B,H,Wf] with channels[k] = 1 (grey, replicated) or 3 channels, its pixels pixel_strides[k] floats apart (a
 * channel slice of a wider NHWC tensor is fine), optionally multiplied by the one-channel float32 image muls[k] (pixels mul_strides[k]
 * floats apart; muls / its entries may be NULL) and by scales[k] (scales may be NULL = 1): test_step_FFHQ's third figure is
 * mask_pred * face * 2 (/root/reference/train_test_GSC.py:872-873).  byte = round-half-even(clamp(v, 0, 1) * 255), each product rounded
 * to float32 as the reference's elementwise operations are.  The pointer arrays are HOST arrays of DEVICE pointers.  out, out_stride,
 * scratch, stream: as bsr_png_encode with W = n_figs * Wf.  ABI 6. */
int bsr_png_encode_figs(int device, int n_figs, const float* const* figs, const float* const* muls, const float* scales, const int* channels,
                        const int* pixel_strides, const int* mul_strides, int B, int H, int Wf, unsigned char* out, size_t out_stride, void* scratch,
                        void* stream);

/* The per-item post-processing of FSRNet.test_step on the device: replaces /root/reference/train_test_GSC.py:424-748 (resize to the crop
 * box + zero pad :437-477, region thresholds :479-590, 4-connected components :594-615, nose rule :650-666, composite :711-722, SSIM /
 * PSNR :724-725, the seven figures of :744 as one strip) for a batch of B items.
 * rows10: [B,S,S,10] float32 = input 3 | ground truth 3 | con_rgb 3 | dif 1 (row 0 of each item's generator call, :419-422);
 * masks: [B,7,S,S] uint8 grey levels of the seven segmentation masks in the order of :386-392 (face+hair, face, mouth, nose, eyebrow,
 * eye, glasses — what cv2.imread returns, before the / 255.0); boxes: [B,4] float32 crop boxes.  All device pointers.
 * losses: [B,2] float32 = ssim, psnr; strips: [B,S,7*S,3] uint8 RGB (the figure strip Logging.save_img writes — feed it to
 * bsr_png_encode); figs: optional [B,7,S,S,3] float32 (the figures themselves; may be NULL); status: [B] int32 — 0 = done,
 * 1 = a mask the reference takes a bounding box of is empty (the reference raises there), 2 = the crop box does not fit S.
 * scratch: bsr_ucb_post_scratch_bytes(B, S) bytes, 256-byte aligned.  S in {32, 64, 128, 256} (reference: 256).  Every decision
 * (rounded masks, thresholds, components, rules) is bit-identical to blindshadowremoval_amd/ucb_post.py, the host statement.  ABI 5. */
size_t bsr_ucb_post_scratch_bytes(int B, int S);
int bsr_ucb_post(int device, const float* rows10, const unsigned char* masks, const float* boxes, int B, int S, float* losses,
                 unsigned char* strips, float* figs, int* status, void* scratch, void* stream);

/* Test hook: the fused NonLocalBlock attention kernel alone (/root/reference/model.py:51-53).
 * qkv [B,tokens,384] (theta | phi | g, 128 channels each) -> y [B,tokens,128]; tokens % 128 == 0. */
int bsr_debug_attention(const float* qkv, float* y, int B, int tokens, void* stream);
/* The same with the kernel of a given BSR_DTYPE_*: F32 = fp32 matrix cores; F32X3 / F16 = the split-precision kernel (attention is
 * split-precision in both 16-bit modes). */
int bsr_debug_attention_dtype(const float* qkv, float* y, int B, int tokens, int dtype, void* stream);
/* The two steps of the F32X3 / F16 case above, separately (ABI 7): the kernel of the 16-bit modes (csrc/attention_h16.h) reads theta | phi | g
 * as the conv3|theta|phi|g GEMM leaves them in those modes — per token 3 x [128 fp16 hi | 128 fp16 lo] (1536 bytes, hi = fp16(x),
 * lo = fp16(x - hi), theta pre-scaled by log2 e).  bsr_debug_split_qkv converts an fp32 [B,tokens,384] tensor to that layout (same size
 * in bytes), bsr_debug_attention_split runs the attention kernel on it; pv1 != 0 = the P.V product with the hi planes only. */
int bsr_debug_split_qkv(const float* qkv, void* qkv_split, int B, int tokens, void* stream);
int bsr_debug_attention_split(const void* qkv_split, float* y, int B, int tokens, int pv1, void* stream);
/* The fp32 kernel with a given workgroup shape: qw = query waves per workgroup (4 = 128 queries, 2 = 64, 1 = 32; 0 = what the
 * forward picks for this batch: the largest block that still gives every CU a workgroup).  All shapes give bit-identical outputs. */
int bsr_debug_attention_qw(const float* qkv, float* y, int B, int tokens, int qw, void* stream);

/* Measurement hook (ABI 7): one wave on `stream` writes (shader cycle counter, 100-MHz real-time counter) pairs to out[2 * samples] every
 * spin x ~3.4 us until *stop (device memory, written from another stream) is non-zero or `samples` pairs are taken; *taken receives the
 * count.  GHz over an interval = d(cycles) / d(ticks) x 0.1.  bench.py runs it beside its `sustained` region: the clock the chip holds
 * under the forward's load, from inside the chip. */
int bsr_clock_trace(int device, unsigned long long* out, int samples, int spin, const int* stop, int* taken, void* stream);

void bsr_destroy(bsr_handle* h);

const char* bsr_last_error(void);

/* ABI version of this header (bumped on any signature change). */
int bsr_abi_version(void);

/* The hash (first 16 hex digits of SHA-256) of the sources this binary was compiled from: every file under csrc/ plus this header,
 * as blindshadowremoval_amd.build.source_sha16() computes it.  The Python binding refuses to load a library whose hash differs from
 * the tree's (a stale built artefact), and bench.py prints it; "unhashed" when compiled outside build.py.  ABI 5. */
const char* bsr_source_sha(void);

#ifdef __cplusplus
}
#endif
#endif /* BSR_HIP_H_ */
