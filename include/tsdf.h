/*
 * tsdf.h -- C ABI of the MI355X-native tracking_sdf hot path (libtsdf_hip.so).
 *
 * One handle = one voxel volume (or one x-slab of it) resident in the HBM of one
 * GPU + the camera pose / intrinsics the reference keeps in CameraTracking.
 * It replaces, for the per-frame hot path only, these reference interfaces
 * (paths relative to the reference's src/):
 *
 *   tsdf_create / tsdf_destroy        SDF::SDF                     include/sdf_3d_reconstruction/sdf.h:78-79, src/sdf.cpp:8-51
 *                                     CameraTracking::CameraTracking  camera_tracking.h:63, src/camera_tracking.cpp:3-18
 *   tsdf_set_intrinsics               CameraTracking::camera_info_cb  src/camera_tracking.cpp:22-36
 *   tsdf_set_camera_transformation    CameraTracking::set_camera_transformation  camera_tracking.h:84, src/camera_tracking.cpp:59-65
 *   tsdf_get_pose                     public fields rot, trans, rot_inv, rot_inv_trans  camera_tracking.h:43-49
 *   tsdf_set_frame / _device          the borrowed cloud_filtered + normals arguments of the two hot calls
 *   tsdf_integrate                    SDF::update                  sdf.h:161-163, src/sdf.cpp:224-315
 *   tsdf_track                        CameraTracking::estimate_new_position  camera_tracking.h:101, src/camera_tracking.cpp:66-245
 *   tsdf_track_and_integrate          the pair of calls in kinect_callback  src/sdf_reconstruction.cpp:69-74
 *   tsdf_accumulate                   one Gauss-Newton pass, src/camera_tracking.cpp:81-189 (+ get_partial_derivative :246-363)
 *   tsdf_gn_update                    src/camera_tracking.cpp:191-239 (+ eigen_utils::direct_exponential_map, src/eigen_utils.cpp:85-128)
 *   tsdf_sample                       SDF::interpolate_distance    sdf.h:86, src/sdf.cpp:127-163
 *   tsdf_download / tsdf_upload       the raw D / W (/Color_W,R,G,B) arrays the visualiser reads  sdf.h:41-55, src/sdf.cpp:47-49
 *   tsdf_mesh_extract / _read         pcl::MarchingCubesSDF::performReconstruction  marching_cubes_sdf.h:434-435, src/marching_cubes_sdf.cpp:243-287
 *                                     + the per-vertex colours of SDF::visualize / SDF::interpolate_color  src/sdf.cpp:353-383, :164-217
 *
 * Conventions
 *   - every call returns int: TSDF_OK (0) or a negative tsdf_status; nothing
 *     throws or aborts across the ABI; tsdf_last_error() gives a message.
 *   - matrices are row-major double[9]; vectors double[3].
 *   - images are organised, row-major, pixel (col,row) at [row*width + col]
 *     (PCL's at(col,row)); xyz / nrm are float[h*w*3], rgb is uint8[h*w*3];
 *     NaN marks invalid depth / undefined normal, exactly as in the reference.
 *   - volume arrays use the reference's linear index  idx = m*m*i + m*j + k
 *     (i = x slowest, k = z fastest; sdf.h:120).
 *   - a handle is single-caller (the reference has one ROS spinner thread).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with TSDF_E_NO_DEVICE / TSDF_E_HIP.
 */
#ifndef TSDF_H_
#define TSDF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: tsdf_config.carry_threads, tsdf_preproc_params.grid_filter (the struct layouts changed: a caller built against
 * version 1 must be rebuilt; tsdf_abi_version() lets a loader refuse a mismatched library)
 * 3: BEHAVIOURAL: planes handed to tsdf_set_frame_device / tsdf_queue_frame_device stay borrowed past the next
 * set_frame* call (their packing runs asynchronously inside the frame's integrate launch); tsdf_device_frame_released()
 * says when they are free.  Version 2 promised "until the next set_frame* call": a caller that alternated two device
 * buffers by that rule must ask tsdf_device_frame_released() (or keep three buffers) -- see tsdf_set_frame_device.
 * 4: tsdf_config.slab_stride (block-cyclic placement of a multi-GPU job's handles; the struct grew from 96 to 104 bytes).
 *    Also since this version, without a change of layout: two frames may wait in the frame queue (tsdf_queue_frame). */
#define TSDF_ABI_VERSION 4

typedef enum tsdf_status {
    TSDF_OK = 0,
    TSDF_E_BADARG = -1,        /* null pointer, bad size, bad slab ...                                 */
    TSDF_E_NO_DEVICE = -2,     /* no HIP device / device ordinal out of range                           */
    TSDF_E_HIP = -3,           /* a HIP runtime call failed (message in tsdf_last_error)                */
    TSDF_E_NO_INTRINSICS = -4, /* tsdf_integrate before tsdf_set_intrinsics (reference: exit(0), sdf.cpp:227-230) */
    TSDF_E_NO_FRAME = -5,      /* hot call before tsdf_set_frame / tsdf_set_frame_device                */
    TSDF_E_SINGULAR = -6,      /* normal matrix singular / pose not finite (reference: silent NaN pose, camera_tracking.cpp:191) */
    TSDF_E_NO_SAMPLES = -7,    /* no valid tracking sample in the frame                                 */
    TSDF_E_HALO = -8,          /* a tracking look-up left this rank's slab+halo: halo too small         */
    TSDF_E_COMM = -9,          /* RCCL / all-reduce hook failure                                        */
    TSDF_E_NOMEM = -10
} tsdf_status;

/* All the constants the reference hard-codes. */
typedef struct tsdf_config {
    int32_t m;                  /* voxels per axis                       sdf_reconstruction.cpp:85 (256) */
    float   width, height, depth; /* extent in metres along x, y, z      (6.0, 6.0, 3.5)                 */
    double  origin[3];          /* world position of the volume corner   (-3,-3,-0.5)                    */
    float   delta;              /* truncation distance                   (0.3)                           */
    float   epsilon;            /* weight plateau                        (0.025)                         */
    int32_t gn_max_iter;        /* Gauss-Newton iterations               sdf_reconstruction.cpp:88 (20)  */
    float   max_twist_diff;     /* signed stop threshold                 (0.001)                         */
    float   v_h;                /* translation step, voxels              (1.0)                           */
    float   w_h;                /* rotation step, radians                (0.01)                          */
    int32_t pixel_stride;       /* tracker sampling stride               camera_tracking.cpp:162-163 (3) */
    int32_t stale_carry;        /* 1 = reproduce the carry-over of camera_tracking.cpp:156-159,261-268   */
    int32_t carry_threads;      /* the OpenMP thread count of the reference run to reproduce (>= 1): the  */
                                /* carry state is thread-local and resets at the first sample of each of  */
                                /* the static-schedule column chunks (camera_tracking.cpp:72-76,146-162); */
                                /* 1 = one thread visits every column (the canonical order)               */
    int32_t with_color;         /* 1 = also keep Color_W,R,G,B           sdf.cpp:294-304                 */
    /* placement: which part of the volume this handle owns, and where it lives */
    int32_t slab_x0, slab_x1;   /* owned x range [x0,x1); 0,m (or 0,0) = whole volume                    */
    int32_t halo;               /* extra x layers kept (and integrated) on each side of the slab         */
    int32_t device;             /* HIP device ordinal                                                    */
    int32_t slab_stride;        /* 0: one slab [x0,x1).  > 0 (ABI 4): BLOCK-CYCLIC placement -- the handle owns the      */
                                /* blocks [x0 + b*stride, x1 + b*stride), b = 0, 1, ... inside the grid, each stored   */
                                /* with its halo: rank r of N takes x0 = r*B, x1 = (r+1)*B, stride = N*B, so that      */
                                /* every rank holds a share of every view (a camera that sweeps across the x axis     */
                                /* leaves a plain slab with 1.7-3.6 x the mean work).  Needs m a power of   */
                                /* two and a multiple of B = x1 - x0, and stride >= B + 2*halo.  tsdf_download /          */
                                /* _upload walk the blocks in increasing x; tsdf_mesh_extract meshes them one by one;    */
                                /* tsdf_save writes every stored layer of every block, tsdf_load takes such a file of    */
                                /* the same placement or any plain file that covers the blocks (a whole-volume one);    */
                                /* only the *_with_halo uploads refuse such a handle (its layers are not one range).    */
} tsdf_config;

typedef struct tsdf_handle tsdf_handle;

typedef struct tsdf_integrate_stats {
    int64_t n_updated;          /* voxels whose D/W were rewritten, owned slab only (halo excluded)      */
    int64_t n_updated_halo;     /* same, halo layers (redundant work, multi-GPU only)                    */
    int64_t n_voxels;           /* voxels swept (slab + halo)                                            */
} tsdf_integrate_stats;

typedef struct tsdf_accum_stats {
    int64_t n_samples;          /* sampled pixels                                                        */
    int64_t n_nan;              /* NaN xyz                                                               */
    int64_t n_oog;              /* centre voxel outside the grid                                         */
    int64_t n_in_grid_owned;    /* in grid and owned by this rank (these do the 13 look-ups)             */
    int64_t n_ok;               /* owned, all 13 look-ups valid                                          */
    int64_t n_terms;            /* owned JJ^T/Jr terms added, stale re-adds included                     */
} tsdf_accum_stats;

typedef struct tsdf_track_stats {
    int32_t iterations;         /* Gauss-Newton iterations run                                           */
    int32_t stopped;            /* 1 = signed stop rule fired (camera_tracking.cpp:216-224)              */
    int64_t n_terms_last;       /* global term count of the last iteration                               */
    double  last_twist[6];
} tsdf_track_stats;

/* Host-provided sum all-reduce over ranks of `n` doubles, in place.  Return 0 on success. */
typedef int (*tsdf_allreduce_fn)(double *buf, int32_t n, void *ctx);

/* ---- lifecycle --------------------------------------------------------------------------- */
int  tsdf_abi_version(void);
void tsdf_default_config(tsdf_config *cfg);                       /* the reference's hard-coded values */
int  tsdf_create(const tsdf_config *cfg, tsdf_handle **out);
void tsdf_destroy(tsdf_handle *h);
const char *tsdf_last_error(const tsdf_handle *h);               /* h may be NULL: last create() error */
const char *tsdf_strerror(int status);
int  tsdf_get_config(const tsdf_handle *h, tsdf_config *cfg);

/* ---- camera state ------------------------------------------------------------------------ */
int tsdf_set_intrinsics(tsdf_handle *h, const double K[9]);
int tsdf_set_camera_transformation(tsdf_handle *h, const double rot[9], const double trans[3]);
/* The four constants of CameraTracking::CameraTracking (camera_tracking.cpp:3-18, definition order: iterations,
 * signed stop threshold, v_h in voxels, w_h in radians) after creation: the reference takes them in the tracker's
 * constructor, i.e. after the SDF exists.  On a sharded volume a larger w_h needs a larger halo (TSDF_E_HALO says so). */
int tsdf_set_tracker_params(tsdf_handle *h, int32_t gn_max_iter, float max_twist_diff, float v_h, float w_h);
int tsdf_get_pose(const tsdf_handle *h, double rot[9], double trans[3],
                  double rot_inv[9], double rot_inv_trans[3]);    /* any pointer may be NULL */

/* ---- per-frame input ------------------------------------------------------------------------
 * tsdf_set_frame copies host images to the device (through pinned staging; page-locked caller buffers -- hipHostMalloc /
 * hipHostRegister -- are copied from directly; either way the buffers are only borrowed for the call).  nrm / rgb may be NULL
 * (tracking needs xyz only; integration needs nrm; rgb is needed only when with_color = 1).
 * tsdf_set_frame_device borrows DEVICE pointers (same layouts) whose contents must be complete when the call is made.
 * Nothing is launched by the call: the tracker's first pass reads its samples from the xyz plane and the pixel records
 * are packed inside the frame's integrate launch (workgroups appended to its first kernel, asynchronous).
 * BORROWING RULE (ABI version 3): the planes of the frame with serial s (tsdf_frame_serial() right after the call) must
 * stay valid and unchanged until tsdf_device_frame_released() >= s, or tsdf_synchronize / tsdf_destroy.  In the loop
 * set_frame_device(k) -> track -> integrate(k) -> set_frame_device(k+1) -> ... frame k becomes free a few microseconds
 * into integrate(k)'s launch -- i.e. usually AFTER set_frame_device(k+1) has returned, and certainly by the time the
 * first tsdf_track of frame k+1 has returned.  A producer that refills buffers on its own stream needs a ring of THREE
 * device buffers to never wait (frame k+2 is written while frame k may still be read and frame k+1 is current), or two
 * buffers and a poll of tsdf_device_frame_released() before each refill.  The reference's own contract for comparison:
 * clouds are borrowed for the duration of estimate_new_position / update only (sdf.h:161-163).
 * (TSDF_DEFER_PACK=0 in the environment of tsdf_create: packed by a launch of its own when the frame is set; the rule
 * and tsdf_device_frame_released() are the same.)
 * The frame-side work of tsdf_set_frame / tsdf_set_depth_frame (staging copies, pre-processing, the packing
 * kernel) runs on an internal second stream, so setting frame k+1 right after tsdf_integrate(k) overlaps it
 * with that integration; the hot calls wait for it on the device. */
int tsdf_set_frame(tsdf_handle *h, const float *xyz, const float *nrm, const uint8_t *rgb,
                   int32_t width, int32_t height);
int tsdf_set_frame_device(tsdf_handle *h, const float *d_xyz, const float *d_nrm, const uint8_t *d_rgb,
                          int32_t width, int32_t height);
/* Serial (as tsdf_frame_serial counts; a queued frame has the serial it will get from tsdf_next_frame) of the newest
 * frame such that the library no longer reads the DEVICE planes of that frame or of any device frame before it.  Host
 * frames are borrowed for their call only and never hold the value back.  Non-blocking, launches nothing: the launch
 * that packs a device frame stores a ticket in pinned host memory when the planes have been read, and this call
 * compares.  -1 for a NULL handle.  Monotonic; equals tsdf_frame_serial() (+1 with a packed queued device frame) once
 * everything handed over has been read, e.g. after tsdf_synchronize. */
int64_t tsdf_device_frame_released(const tsdf_handle *h);
/* Frame queue: upload coming frames while the current one is still being tracked and integrated
 * (the reference's callback has the next cloud in its subscriber queue while it works on the current one,
 * sdf_reconstruction.cpp:89: queue size 1).  tsdf_queue_frame / tsdf_queue_frame_aos return at once: page-locked
 * plane buffers are read by the DMA engine, pageable ones (and PCL-style arrays of structs) by a library thread that
 * repacks them into pinned staging planes with the TSDF_HOST_THREADS pool -- either way the caller's buffers are
 * BORROWED UNTIL THE tsdf_next_frame THAT MAKES THE FRAME CURRENT RETURNS and must not change meanwhile.  The hot
 * calls in between (tsdf_track, tsdf_integrate, tsdf_track_and_integrate) keep working on the current frame.
 * tsdf_next_frame makes the oldest queued frame the current one (device-side wait, no host block beyond the end of the
 * host-side repack).  TWO frames can wait behind the current one (one, until round 5 of this library): a frame's staging + copy
 * (the copy alone is 155 us for 640x480, a frame's GPU work 190 us) then has two frame times instead of one, which is
 * what takes the queue from 0.85-0.93 of the device-resident rate to 0.95+.  A third tsdf_queue_frame* is
 * TSDF_E_BADARG; so is a frame of another size than the current / queued ones, tsdf_set_frame* while a frame is queued,
 * and tsdf_queue_frame_device behind a queued frame (a frame in device memory has nothing to hide: first place only).
 * tsdf_queue_frame_aos needs the points (it has no 'normals only' form). */
int tsdf_queue_frame(tsdf_handle *h, const float *xyz, const float *nrm, const uint8_t *rgb, int32_t width, int32_t height);
/* The same for a frame that is already in device memory (tsdf_set_frame_device's layouts and borrowing rule: the
 * buffers must be complete when the call is made and stay valid and unchanged until tsdf_device_frame_released()
 * reaches the serial the frame gets from tsdf_next_frame -- normally a few microseconds into the CURRENT frame's
 * integrate launch, which packs it -- or tsdf_synchronize).  Only the packing is left to hide: the integrate launch of the CURRENT frame
 * does it, sample list included, in workgroups appended to its first kernel; a queued frame that no integrate launch
 * came by is taken over unpacked by tsdf_next_frame, exactly like a frame of tsdf_set_frame_device. */
int tsdf_queue_frame_device(tsdf_handle *h, const float *d_xyz, const float *d_nrm, const uint8_t *d_rgb, int32_t width, int32_t height);
int tsdf_next_frame(tsdf_handle *h);
/* Number of frames made current so far (every successful tsdf_set_frame* / tsdf_set_depth_frame adds one; -1 for a
 * NULL handle): lets a caller that uploads a cloud for estimate_new_position check, at SDF::update time
 * (sdf_reconstruction.cpp:70,74), that the frame in the library is still that upload. */
int64_t tsdf_frame_serial(const tsdf_handle *h);

/* The frame in the reference's OWN format: arrays of point structs, as pcl::PointCloud<pcl::PointXYZRGB>::points and
 * pcl::PointCloud<pcl::Normal>::points hold them (sdf_reconstruction.cpp:33-49; PCL pads both to 32 bytes) -- any
 * array-of-structs layout is described by strides and byte offsets.  A few library threads repack it straight into
 * the pinned staging planes (one pass over the caller's memory, no intermediate copy).  The reference hands the same
 * cloud to estimate_new_position (points only) and then to update (points + normals): pass points = NULL in the
 * second call to keep the xyz / rgb uploaded by the first and add only the normals (TSDF_E_NO_FRAME unless the current
 * frame came from a host buffer of the same size).  Host buffers are borrowed for the call only. */
typedef struct tsdf_aos_layout {
    int32_t point_stride;                   /* bytes from one point to the next                         */
    int32_t xyz_offset;                     /* x, y, z: three consecutive floats at this byte offset    */
    int32_t r_offset, g_offset, b_offset;   /* one byte each; any of them < 0 = the points have no colour */
    int32_t normal_stride, normal_offset;   /* same for the normals: three consecutive floats           */
} tsdf_aos_layout;
int tsdf_set_frame_aos(tsdf_handle *h, const void *points, const void *normals, const tsdf_aos_layout *layout,
                       int32_t width, int32_t height);
int tsdf_queue_frame_aos(tsdf_handle *h, const void *points, const void *normals, const tsdf_aos_layout *layout,
                         int32_t width, int32_t height);   /* see tsdf_queue_frame */

/* The reference's two hot calls on its own clouds, as kinect_callback issues them (sdf_reconstruction.cpp:70,74):
 *   tsdf_track_aos      = CameraTracking::estimate_new_position(sdf, cloud)        camera_tracking.h:101
 *   tsdf_integrate_aos  = SDF::update(tracker, cloud, normals)                     sdf.h:161-163
 * tsdf_track_aos makes `points` the current frame and tracks it: the tracker's samples (every pixel_stride-th point of
 * every pixel_stride-th row, 0.5 MB at 640x480) are gathered and copied first and the Gauss-Newton passes run on them while
 * the library's threads repack the whole cloud and the frame stream copies it -- the 4.6 MB of a 640x480 cloud travel
 * UNDER the passes instead of in front of them.  The cloud is borrowed for the call only.  Pose / error behaviour as
 * tsdf_track.  The frame has no normals yet: tsdf_integrate needs tsdf_integrate_aos (or tsdf_set_frame_aos) first.
 * tsdf_integrate_aos completes the frame with `normals` and integrates it.  `points` may be NULL (= the tracked cloud)
 * or the cloud itself, as SDF::update receives it: the library then compares it, every point, with what it staged when
 * the cloud was tracked (on its threads, under the copy of the normals) and uploads it again only when it is another
 * cloud or was changed in place -- SDF::update integrates the cloud as it is when update is called (sdf.cpp:258-259).
 * Without a cloud tracked by tsdf_track_aos (frame 1 of a sequence, sdf_reconstruction.cpp:69) it is
 * tsdf_set_frame_aos(points, normals) + tsdf_integrate.  Buffers are borrowed for the call only. */
int tsdf_track_aos(tsdf_handle *h, const void *points, const tsdf_aos_layout *layout, int32_t width, int32_t height,
                   tsdf_track_stats *stats);
/* The same with the frame's normals handed over at tracking time -- one more argument than the reference's
 * estimate_new_position has, for a caller that can give it: kinect_callback holds `normals` before it tracks
 * (sdf_reconstruction.cpp:44-49,70).  The whole frame (8.3 MB at 640x480) is then staged and copied under the
 * Gauss-Newton passes, and its pixel records are packed behind the copy: when the call returns the frame is complete and
 * tsdf_integrate (no further upload) integrates it -- the synchronous callback then runs at about the rate of the two-deep
 * queue.  Both clouds are borrowed for the call only.  tsdf_integrate_aos after this call still compares BOTH clouds with
 * what was staged, every point and normal, and uploads what differs; tsdf_integrate takes the staged frame as it is. */
int tsdf_track_frame_aos(tsdf_handle *h, const void *points, const void *normals, const tsdf_aos_layout *layout,
                         int32_t width, int32_t height, tsdf_track_stats *stats);
int tsdf_integrate_aos(tsdf_handle *h, const void *points, const void *normals, const tsdf_aos_layout *layout,
                       int32_t width, int32_t height, tsdf_integrate_stats *stats);

/* ---- optional: depth pre-processing on the GPU.  Replaces, for callers that have a raw
 * depth image instead of PCL clouds, the host-side steps of sdf_reconstruction.cpp:29-49 (cloud conversion,
 * pcl::FastBilateralFilter, pcl::IntegralImageNormalEstimation).  PCL is not available here, so this is the
 * repository's own stand-in (specified in csrc/preproc_kernels.hip and tests/preproc_ref.py), NOT a bit-level
 * reimplementation of PCL: parity with PCL is unpinned.  The result becomes the current frame exactly as if
 * tsdf_set_frame had been called with the produced xyz / normals / rgb. */
typedef struct tsdf_preproc_params {
    float   depth_scale;        /* metres per unit of a uint16 depth image (TUM: 1/5000); 0 = invalid pixel     */
    float   sigma_s;            /* bilateral spatial sigma, pixels (PCL default 15)                               */
    float   sigma_r;            /* bilateral range sigma, metres   (PCL default 0.05)                             */
    int32_t radius;             /* bilateral window radius, pixels; 0 = no filtering; <= 32 (default 2*sigma_s capped) */
    int32_t normal_radius;      /* gradient averaging radius, pixels (reference smoothing size 10 -> 5); 1..8      */
    float   max_depth_change;   /* depth-discontinuity factor (reference 0.02)                                    */
    int32_t grid_filter;        /* 1 (default): bilateral GRID (Paris & Durand; the algorithm family of pcl::FastBilateralFilter),
                                   sigma_s in [1, 30], `radius` only switches it off (0); 0: exact windowed filter of `radius` */
} tsdf_preproc_params;
void tsdf_default_preproc(tsdf_preproc_params *p);
/* depth16 (uint16) or depthf (float metres, <= 0 / NaN invalid): exactly one non-null; host pointers; rgb may be null */
int tsdf_set_depth_frame(tsdf_handle *h, const uint16_t *depth16, const float *depthf, const uint8_t *rgb,
                         int32_t width, int32_t height, const tsdf_preproc_params *params);
/* The same through the frame queue (tsdf_queue_frame's rules: up to two frames waiting, of the current frame's size,
 * buffers borrowed until the tsdf_next_frame that makes the frame current returns): upload and pre-processing -- including its one host round
 * trip for the bilateral grid's depth range -- run on a library thread and the frame stream while the caller tracks and
 * integrates the current frame (the pixel records are written by the frame's own integrate launch).  What
 * tsdf_set_depth_frame would have returned for the frame (bad depth range ...) is returned by tsdf_next_frame. */
int tsdf_queue_depth_frame(tsdf_handle *h, const uint16_t *depth16, const float *depthf, const uint8_t *rgb,
                         int32_t width, int32_t height, const tsdf_preproc_params *params);
/* copy the current frame's planes back as the library holds them (any pointer may be null): xyz, normals as float[h*w*3];
 * for frames that came from host memory or as raw depth (also while the next frame is queued behind them), TSDF_E_NO_FRAME
 * for frames handed over in device memory */
int tsdf_get_preprocessed(tsdf_handle *h, float *xyz, float *nrm);

/* ---- the hot path ------------------------------------------------------------------------- */
int tsdf_integrate(tsdf_handle *h, tsdf_integrate_stats *stats);  /* SDF::update at the current pose */
/* estimate_new_position: updates the pose.  On ANY error (also one raised by a later Gauss-Newton pass, after earlier
 * passes have moved the pose) the handle's pose is the one it had when the call was made. */
int tsdf_track(tsdf_handle *h, tsdf_track_stats *stats);
/* The two hot calls of kinect_callback back to back (sdf_reconstruction.cpp:69-74): track when do_track != 0 (every
 * frame but the first), then integrate at the resulting pose.  One ABI crossing instead of two: the integration is
 * launched the moment the last Gauss-Newton pass is solved.  A tracking error is returned and nothing is integrated. */
int tsdf_track_and_integrate(tsdf_handle *h, int32_t do_track, tsdf_track_stats *track_stats, tsdf_integrate_stats *integrate_stats);
/* One accumulation pass at the current pose.  A (6x6 row-major) and b are this rank's partial sums
 * (NOT all-reduced), so a test can add the partials of several slabs itself. */
int tsdf_accumulate(tsdf_handle *h, double A[36], double b[6], tsdf_accum_stats *stats);
/* Solve + exponential map + stop rule + pose update for given (already reduced) A, b.  Host only.
 * *stop receives the stop-rule result.  On TSDF_E_SINGULAR the pose is left unchanged. */
int tsdf_gn_update(tsdf_handle *h, const double A[36], const double b[6], double twist[6], int32_t *stop);
/* Batched SDF::interpolate_distance on the device: vox = n x 3 continuous voxel coordinates (host). */
int tsdf_sample(tsdf_handle *h, const double *vox, int32_t n, float *val, int32_t *ok);

/* ---- volume I/O (host mirrors for meshing / checkpoints), reference index order, owned slab only:
 *      arrays hold (slab_x1-slab_x0)*m*m floats starting at voxel (slab_x0,0,0). */
int tsdf_download(tsdf_handle *h, float *D, float *W);
int tsdf_upload(tsdf_handle *h, const float *D, const float *W);
int tsdf_download_color(tsdf_handle *h, float *Color_W, float *R, float *G, float *B);
int tsdf_upload_color(tsdf_handle *h, const float *Color_W, const float *R, const float *G, const float *B);
/* Re-integrate nothing, just make halo layers consistent after tsdf_upload on a sharded volume:
 * uploads D/W (colour) for the halo layers too (arrays cover [max(0,x0-halo), min(m,x1+halo)) ). */
int tsdf_upload_with_halo(tsdf_handle *h, const float *D, const float *W);
int tsdf_upload_color_with_halo(tsdf_handle *h, const float *Color_W, const float *R, const float *G, const float *B);
int tsdf_reset(tsdf_handle *h);                                   /* back to the constructor state */
/* Volume checkpoint (the reference keeps the volume only in RAM, sdf.cpp:52-54).
 * File = 80-byte little-endian header {"TSDFVOL2", int32 m, x0, x1, has_color, float width, height, depth, delta,
 * epsilon, int32 xs, double origin[3], int32 xe, int32 reserved} followed by D, W (and Color_W, R, G, B) float arrays
 * of the x layers [xs, xe) = the writer's slab AND its halo, in reference index order.
 * tsdf_load restores every layer the handle stores (slab + halo) from a file that covers them -- the same shard's
 * file, or the file of a whole (unsharded) volume -- so the halo layers of a restored shard are again bit-identical
 * to the neighbour's interior.  It refuses (TSDF_E_BADARG) a file written for another m / extent / origin / delta /
 * epsilon / colour setting and (TSDF_E_HALO) one that does not cover the stored layers.  Only the voxel state is
 * restored: pose and intrinsics stay with the caller. */
int tsdf_save(tsdf_handle *h, const char *path);
int tsdf_load(tsdf_handle *h, const char *path);

/* ---- mesh extraction (the reference's visualiser thread, sdf.cpp:317-391) ---------
 * tsdf_mesh_extract runs marching cubes over the cubes whose base voxel this handle owns (interior voxels
 * 1..m-2 only, sdf.cpp:36-39; gate: all eight corner weights > 0, marching_cubes_sdf.cpp:219-239) and keeps the
 * triangle soup in HBM: 9 floats per triangle, cubes in reference index order, vertex positions computed in
 * float exactly as createSurface does (grid-local frame: add cfg.origin for world coordinates, sdf.cpp:355-369;
 * the reference's half-voxel offset of that frame is kept).  with_color != 0 also evaluates
 * SDF::interpolate_color at every vertex's world position (4 floats r,g,b,a per vertex, a = 1; r,g,b carry
 * the reference's scaling: /255 when interpolated, raw 0..255 on an exact voxel hit, NaN with no coloured
 * corner).  iso_level must be in [0,1) as in the reference (marching_cubes_sdf.cpp:246-252; it uses 0).
 * The case table is the reference's (marching_cubes_sdf.h:73-364, kept as data: tools/gen_mc_tables.py), so the soup is
 * performReconstruction's, triangle for triangle and in its order.
 * On a sharded volume each rank meshes its own cubes (needs halo >= 1); rank order = index order.
 * tsdf_mesh_read copies the last extraction to host arrays (colors may be NULL); tsdf_mesh_device hands out
 * the device buffers (valid until the next extraction / destroy). */
int tsdf_mesh_extract(tsdf_handle *h, float iso_level, int32_t with_color, int64_t *n_triangles);
int tsdf_mesh_read(tsdf_handle *h, float *vertices /* n*9 */, float *colors /* n*12 or NULL */, int64_t capacity_triangles);
int tsdf_mesh_device(tsdf_handle *h, const float **vertices, const float **colors, int64_t *n_triangles);

/* ---- multi-GPU (one process per GPU; the volume is sharded in x-slabs) ------------------- */
/* Owned range of `rank` out of `nranks` for an m-voxel axis (balanced contiguous slabs). */
int tsdf_slab_range(int32_t m, int32_t nranks, int32_t rank, int32_t *x0, int32_t *x1);
/* Block-cyclic placement for tsdf_config::slab_x0 / slab_x1 / slab_stride: rank r of nranks owns the blocks
 * [r B + j nranks B, (r+1) B + j nranks B).  block = 0 picks B = m / (2 nranks) rounded down to a power of two (two blocks per
 * rank), doubled until the stored ranges of a rank's blocks cannot overlap (nranks B >= B + 2 halo).  TSDF_E_BADARG when
 * nothing fits (m must be a power of two).  Every rank holds a share of every view: on a path that sweeps the camera across
 * x the busiest of 8 ranks integrates 20-37 % faster than with the best static slabs. */
int tsdf_cyclic_range(int32_t m, int32_t nranks, int32_t rank, int32_t halo, int32_t block, int32_t *x0, int32_t *x1, int32_t *stride);
/* Slabs of equal WORK instead of equal thickness.  A camera frustum covers the middle of the volume: with equal slabs the
 * 8-way split of config 5 gives the busiest rank 4.5 x the average work (profiles/r05_rank_costs_*), and every Gauss-Newton
 * pass and every integration waits for that rank.  layer_weight[i] >= 0 is the expected work of x layer i (m entries, e.g.
 * from tsdf_frustum_layer_weights); a rank pays for the layers it STORES (slab + halo per side); the boundaries minimise the
 * largest such sum.  Deterministic: every rank computes the same cuts from the same weights.  All-zero weights give
 * tsdf_slab_range's equal slabs. */
int tsdf_slab_range_weighted(int32_t m, int32_t nranks, int32_t rank, int32_t halo, const double *layer_weight,
                             int32_t *x0, int32_t *x1);
/* Expected integration work per x layer (in 64-voxel work items) for a camera at (rot, trans) with intrinsics K and a
 * width x height image: the layer's voxels inside the view frustum up to max_depth metres, plus a floor per stored layer.
 * ADDS to weights[0..m), so several poses (the initial one; a planned path) accumulate.  Host only, no handle. */
int tsdf_frustum_layer_weights(const tsdf_config *cfg, const double K[9], int32_t width, int32_t height, const double rot[9],
                               const double trans[3], float max_depth, double *weights);
/* Halo (x layers per side) that covers every tracking look-up of points up to max_range metres
 * from the world origin: ceil(w_h * max_range * m/width) + ceil(v_h) + 2. */
int32_t tsdf_halo_for(const tsdf_config *cfg, float max_range);
/* RCCL path: rank 0 calls tsdf_comm_unique_id, the 128 bytes are broadcast by the host
 * (torch.distributed / MPI / a file), every rank calls tsdf_comm_init. */
int tsdf_comm_unique_id(void *id128);
int tsdf_comm_init(tsdf_handle *h, int32_t nranks, int32_t rank, const void *id128);
/* Alternative for ranks on ONE node: every rank stores its 34-double row + {generation, pass number} into its slot of
 * a POSIX shared-memory segment and every rank's host sums the slots in rank order.  No GPU collective, no stream
 * synchronisation, bitwise deterministic.  The call is a rendezvous (bounded by 20 s): rank 0 replaces whatever
 * carries `name`, creates and zero-fills the segment and picks a generation; the others attach and join; once all
 * have joined rank 0 unlinks the name, so nothing is left behind in /dev/shm even if the job dies later.  `name`
 * must be unique among jobs that initialise at the same time (e.g. contain the master port). */
int tsdf_comm_init_shm(tsdf_handle *h, int32_t nranks, int32_t rank, const char *name);
/* Device-side alternative for ranks on ONE node (at most 64): no host in the exchange step.  Every rank owns a small
 * buffer in uncached device memory that every other rank maps through a HIP IPC handle (hipIpcGetMemHandle /
 * hipIpcOpenMemHandle; over xGMI between GPUs, and two ranks may also share one GPU).  The tracker workgroup that
 * finishes a rank's 34-double row stores it into its slot of EVERY rank's buffer, releases {generation, pass number}
 * behind it at system scope, waits for the words of the other ranks in its own buffer and adds the rows in rank
 * order: the same bits on every rank and as the shared-memory fan-in.  A rank that waits longer than 5 s gives up and
 * the call that launched the pass returns TSDF_E_COMM.  Rendezvous, `name` and time limit as tsdf_comm_init_shm (the
 * segment carries the IPC handles); TSDF_E_COMM when the runtime refuses IPC, and nothing is left configured then.
 * Two handles of ONE process (same pid and process token) lend each other their raw buffer pointers instead of IPC
 * mappings: such sibling handles must leave the exchange (tsdf_comm_finalize / tsdf_destroy) before any of them is
 * destroyed. */
int tsdf_comm_init_peer(tsdf_handle *h, int32_t nranks, int32_t rank, const char *name);
int tsdf_comm_finalize(tsdf_handle *h);                          /* drop the RCCL communicator / the shared segment / the peer mappings (hook, if any, takes over) */
/* Alternative: let the host do the 28-double sum (e.g. torch.distributed); fn = NULL removes it. */
int tsdf_set_allreduce_hook(tsdf_handle *h, tsdf_allreduce_fn fn, void *ctx);
/* Sum-all-reduce n doubles through whichever of the two is configured (identity if neither). */
int tsdf_allreduce(tsdf_handle *h, double *buf, int32_t n);

/* ---- host-side algebra of the tracker, usable without a handle or a GPU (it runs on the host in the
 *      product too: 27 doubles per Gauss-Newton pass).  Exposed so that the reference-side caller, the
 *      CPU test-suite and a multi-process driver can use exactly the code tsdf_track uses.
 *   tsdf_host_set_pose             set_camera_transformation: rot_inv = inverse(rot), rot_inv_trans = -rot_inv*trans
 *   tsdf_host_perturbed_rotations  r1p r1m r2p r2m r3p r3m = (I +- w_h [e_k]x) rot          camera_tracking.cpp:92-145
 *   tsdf_host_gn_step              twist = inverse(A) b; [R|t] = exp(twist); stop rule; rot <- R^T rot,
 *                                  trans <- trans - R^T t  (in place).  Returns TSDF_E_SINGULAR and leaves the
 *                                  pose alone if A is singular or the result is not finite.   camera_tracking.cpp:191-239 */
int tsdf_host_set_pose(const double rot[9], const double trans[3], double rot_inv[9], double rot_inv_trans[3]);
int tsdf_host_perturbed_rotations(const double rot[9], float w_h, double rpm[54]);
int tsdf_host_gn_step(double rot[9], double trans[3], const double A[36], const double b[6],
                      float max_twist_diff, double twist[6], int32_t *stop);

/* ---- measurement helpers ------------------------------------------------------------------ */
/* GPU time measured with HIP events recorded on the handle's own stream around each kernel launch,
 * summed since the last reset.  Off by default.  tsdf_set_timing(h, mask): bit 0 = integrate + pack
 * launches (events are read back lazily, no extra synchronisation); bit 1 = every tracker pass (the stop
 * event of each pass must complete before the host continues, which costs a few microseconds per pass);
 * bits 8..15 = sampling period n for the integrate / pack events (0 or 1: every launch; n: every n-th launch of a
 * kind is bracketed -- an event pair per launch costs a 512^3 frame loop about 6 % of its rate). */
typedef struct tsdf_timing {
    double  integrate_ms;       /* integrate_kernel only                                              */
    int64_t integrate_launches;
    double  track_ms;           /* track_kernel (rows + in-launch fan-in), one launch per GN iteration    */
    int64_t track_launches;
    double  pack_ms;            /* per-frame image packing kernel                                      */
    int64_t pack_launches;
} tsdf_timing;
/* Cumulative work counters since the last reset (device counters; reading synchronizes). */
typedef struct tsdf_counters {
    int64_t n_updated;          /* owned voxels rewritten by tsdf_integrate                            */
    int64_t n_updated_halo;
    int64_t n_voxels_swept;
    int64_t integrate_calls;
    int64_t track_calls;
    int64_t track_iterations;   /* Gauss-Newton passes                                                 */
    int64_t track_in_grid;      /* owned in-grid samples over all passes (each does <= 13 look-ups)    */
    int64_t track_terms;
    int64_t integrate_items;    /* 64-voxel work items the row clip produced (each is one 512-byte {D,W} segment) */
    int64_t track_passes_own_queue; /* Gauss-Newton passes the library's own AQL queue took (an option, TSDF_AQL=1; 0 by default, or when
                                 * tsdf_track.hsaco next to the library is missing / not this build's, or no pass >= 1 qualified); ABI 3 */
} tsdf_counters;
int tsdf_set_timing(tsdf_handle *h, int32_t on);
int tsdf_read_timing(tsdf_handle *h, tsdf_timing *out, int32_t reset);
int tsdf_read_counters(tsdf_handle *h, tsdf_counters *out, int32_t reset);
/* Waits for everything the handle has launched (main stream, frame stream, the queue's library thread).  Device
 * frames whose packing is still deferred (tsdf_set_frame_device / tsdf_queue_frame_device) are packed first, by a
 * launch of their own: when the call returns the library no longer reads any borrowed device plane. */
int tsdf_synchronize(tsdf_handle *h);
/* The hipStream_t the hot kernels (track, integrate, mesh) are launched on, as an opaque pointer (for callers
 * that record their own events).  tsdf_synchronize waits for this and for the internal frame stream. */
void *tsdf_stream(tsdf_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_H_ */
