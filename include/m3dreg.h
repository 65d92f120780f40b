/*
 * m3dreg.h — C ABI of libm3dreg.so, the MI355X (gfx950) scan-registration engine that fills the
 * `gpu_6dslam_node` slot of the m3d pipeline.
 *
 * What this boundary replaces in the reference (all paths relative to /root/reference):
 *   - The reference has NO function-level API for this path: `gpu_6dslam/` is an empty, un-vendored
 *     git submodule (.gitmodules:1-3) and the only coupling is a ROS1 process boundary —
 *     m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13 starts `gpu_6dslam_node`, which
 *     consumes the `sensor_msgs/PointCloud2` that m3d_aggregator publishes
 *     (m3d/m3d_aggregator/src/m3d_aggregator.cpp:188-212, advertise at :174).
 *   - Therefore every entry point below takes the raw PointCloud2 payload (`data` pointer,
 *     `point_step`, byte offsets of the FLOAT32 x/y/z fields — the layout pcl::toPCLPointCloud2
 *     produces for pcl::PointXYZ at m3d_aggregator.cpp:196-201: step 16, x@0 y@4 z@8) and returns a
 *     column-major float[16] pose (Eigen::Matrix4f storage order), so a ROS shim needs no copies
 *     or conversions (see INTEGRATION.md and ros/gpu_6dslam_node.cpp).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary, every function
 *     returns an int status (0 = M3DREG_OK, <0 = m3dreg_error).
 *   - Threading: one handle = one device + one HIP stream + the state of one batch; a handle is not thread-safe, different handles are
 *     independent. No global mutable state shapes a result or a schedule (the only process-wide word is a creation counter that places a
 *     handle's pairs on the XCDs), and (ABI 8) no per-handle history does either: what a registration launches is a function of the batch —
 *     its sizes, levels, whether its targets have tiles, and, for the dense-level search kernel, the clouds' own voxel counts where the host
 *     has read them back (every synchronous creation call does); for clouds it has not, both search kernels are launched and the device
 *     decides pair by pair (an empty launch costs ~5 us, never a bit).
 *   - Poses map SOURCE-frame points into the TARGET frame: p_target = T * p_source.
 *   - All results are bit-reproducible: they do not depend on launch geometry, scheduling or
 *     atomics order (integer fixed-point normal-equation sums).
 *   - There is no CPU fallback: without a usable HIP device m3dreg_create fails with
 *     M3DREG_ERR_NO_DEVICE.
 */
#ifndef M3DREG_H
#define M3DREG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3DREG_ABI_VERSION 8   /* 2: + m3dreg_cloud_create_batch_async, m3dreg_cloud_status, M3DREG_BAD_CLOUD, m3dreg_cloud_desc.source_only,
                                     M3DREG_CLOUD_* flags; m3dreg_align_batch_async refuses a second pending batch
                                  3: + m3dreg_multi_* (one process, several devices), M3DREG_ERR_OUT_OF_MEMORY (every entry point is
                                     exception-guarded), M3DREG_PROFILE_BUCKETING / _REDUCE_SOLVE, cloud lifetime rules (below),
                                     m3dreg_debug_accumulate / _trace refuse to run while a batch is pending
                                  4: + m3dreg_host_alloc / _free / _register / _unregister (pinned payloads); m3dreg_multi_align runs one host thread
                                     per device; m3dreg_default_params is a coarse-to-fine pyramid (see there)
                                  5: + m3dreg_pair_desc.target_group (shared targets are co-located and bucketed once), m3dagg_set_scan_trig,
                                     m3dreg_cloud_density; the schedule no longer depends on what else the process has in flight
                                  6: m3dreg_pair_desc.reserved must be 0 (+ m3dreg_set_batch_chains, removed again in 8)
                                  7: + m3dreg_set_latency_mode (a serial caller states that its batches have the GPU to themselves)
                                  8: + m3dloop_* (loop-closure candidate generation), m3dagg_set_rearm, m3dreg_debug_checks, m3dreg_debug_cloud_raw, m3dreg_profile_batches; - m3dreg_set_batch_chains (internal launch chains
                                     lost on every workload measured); the dense-level schedule is decided per batch, no longer from the handle's previous batch */
#define M3DREG_MAX_LEVELS 4
#define M3DREG_NSUMS 29 /* 21 upper-tri JtJ + 6 Jtr + sum r^2 + correspondence count */

typedef enum m3dreg_error {
    M3DREG_OK = 0,
    M3DREG_ERR_INVALID_ARG = -1,
    M3DREG_ERR_NO_DEVICE = -2,      /* no HIP device / extension unusable: never falls back to CPU */
    M3DREG_ERR_HIP = -3,            /* a HIP runtime call failed (see m3dreg_last_error) */
    M3DREG_ERR_GRID_TOO_LARGE = -4, /* voxel key needs more than 31 bits: coarsen leaf or crop cloud */
    M3DREG_ERR_EMPTY_CLOUD = -5,    /* no finite point in the cloud */
    M3DREG_ERR_NO_TARGET = -6,      /* m3dreg_align before m3dreg_set_target_xyz */
    M3DREG_ERR_LEVEL_MISMATCH = -7, /* cloud was bucketed with other leaf sizes than the handle's */
    M3DREG_ERR_OUT_OF_MEMORY = -8   /* a host allocation failed inside the library (std::bad_alloc caught at the boundary); the handle stays usable */
} m3dreg_error;

typedef enum m3dreg_metric {
    M3DREG_POINT_TO_POINT = 0,
    M3DREG_POINT_TO_PLANE = 1
} m3dreg_metric;

/* Per-registration termination status (m3dreg_stats.status). */
typedef enum m3dreg_status {
    M3DREG_CONVERGED = 0,       /* |d_rot| < eps_rot and |d_trans| < eps_trans at the finest level */
    M3DREG_MAX_ITERATIONS = 1,  /* ran all iterations (normal for fixed-iteration runs) */
    M3DREG_TOO_FEW_CORR = 2,    /* fewer than min_correspondences matches: pose left at last good value */
    M3DREG_RANK_DEFICIENT = 3,  /* 6x6 normal matrix had a pivot <= pivot_rel_tol * max diagonal */
    M3DREG_DIVERGED = 4,        /* update rotation > 2 rad or non-finite */
    M3DREG_BAD_CLOUD = 5        /* a cloud of the pair came out of m3dreg_cloud_create_batch_async in error (no finite point, grid too
                                   large: m3dreg_cloud_status says which); the pose is the initial guess, no iteration ran */
} m3dreg_status;

/*
 * Registration parameters. `gpu_6dslam_node` is launched with no params in the reference
 * (m3d_husky_bringup.launch:13), so all of these are this library's own; m3dreg_default_params
 * fills the defaults quoted in DESIGN.md. Multi-resolution: level 0 is the coarsest and is run
 * first; each level has its own voxel leaf and iteration budget.
 */
typedef struct m3dreg_params {
    int32_t n_levels;                        /* 1..M3DREG_MAX_LEVELS */
    float leaf[M3DREG_MAX_LEVELS];           /* voxel edge [m]; the NN search covers the 27 cells around the query */
    int32_t iterations[M3DREG_MAX_LEVELS];   /* max Gauss-Newton iterations per level */
    float max_corr_dist[M3DREG_MAX_LEVELS];  /* reject a match when d^2 > max_corr_dist^2 */
    int32_t metric;                          /* m3dreg_metric, used at every level */
    int32_t min_correspondences;             /* below this -> M3DREG_TOO_FEW_CORR */
    double eps_rot;                          /* [rad]  convergence threshold on |omega| */
    double eps_trans;                        /* [m]    convergence threshold on |v| */
    double pivot_rel_tol;                    /* LDL^T pivot threshold relative to max diagonal */
    float plane_ratio;                       /* normal valid iff lambda3 <= plane_ratio * lambda2 */
    int32_t normal_min_pts;                  /* normal valid iff >= this many points in the 27 cells */
    float normal_leaf;                       /* voxel edge of the dedicated normal-estimation grid [m] */
    float normal_min_spread;                 /* normal valid iff sqrt(lambda2) >= this * normal_leaf */
} m3dreg_params;

typedef struct m3dreg_stats {
    int32_t status;       /* m3dreg_status */
    int32_t iterations;   /* Gauss-Newton iterations actually executed, all levels */
    int64_t n_corr;       /* correspondences used by the last executed iteration */
    double rms;           /* sqrt(sum r^2 / n_corr) of the last executed iteration (before its update) */
    double last_rot;      /* |omega| of the last update [rad] */
    double last_trans;    /* |v| of the last update [m] */
} m3dreg_stats;

typedef struct m3dreg_handle m3dreg_handle; /* opaque: device, stream, params, workspaces */
typedef struct m3dreg_cloud m3dreg_cloud;   /* opaque: one bucketed cloud resident in HBM */

/* One entry of a batch: both clouds already bucketed and resident on the handle's device. */
typedef struct m3dreg_pair {
    const m3dreg_cloud* source;
    const m3dreg_cloud* target;
    float init_T[16]; /* column-major initial guess, source -> target */
} m3dreg_pair;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int m3dreg_default_params(m3dreg_params* out);
/* `stream`: a hipStream_t passed as void* (NULL = the library creates its own non-blocking stream). */
int m3dreg_create(const m3dreg_params* params, int device, void* stream, m3dreg_handle** out);
int m3dreg_destroy(m3dreg_handle* h);
const char* m3dreg_backend_name(void);            /* "hip-gfx950" */
const char* m3dreg_last_error(const m3dreg_handle* h); /* text of the last failure on this handle */
int m3dreg_abi_version(void);

/* ---- the gpu_6dslam_node call surface: PointCloud2 bytes in, 4x4 pose out ------------------- */
/* Copies the cloud to the device, buckets it (all levels) and, for point-to-plane, estimates normals.
 * `data` is the PointCloud2 `data` buffer (host memory, caller keeps ownership), `n` = width*height. */
int m3dreg_set_target_xyz(m3dreg_handle* h, const void* data, size_t n, size_t point_step,
                          size_t off_x, size_t off_y, size_t off_z);
/* Registers the source cloud against the current target. init_T/out_T: column-major float[16]. */
int m3dreg_align(m3dreg_handle* h, const void* src, size_t n, size_t point_step, size_t off_x,
                 size_t off_y, size_t off_z, const float init_T[16], float out_T[16],
                 m3dreg_stats* stats);

/* ---- resident clouds (loop-closure batches, scan-to-scan chains, benchmarks) ----------------
 * Ownership and lifetime. A cloud belongs to the handle that bucketed it (its OWNER): its device block comes from the owner's pool
 * and returns there, whichever handle is passed to m3dreg_cloud_destroy. A cloud may be used by ANY handle of the same device
 * (source or target of its registrations, m3dmap_insert): the using handle's stream waits for the owner's bucketing, and the
 * owner's stream waits for the last such use before the block is handed out again, so a cloud may be destroyed as soon as the
 * call that uses it has returned (also the enqueue-only m3dreg_align_batch_async). An owner handle outlives its clouds:
 * m3dreg_destroy only marks it closed while clouds of it are alive; the last m3dreg_cloud_destroy then releases it. */
/* `data_is_device`: 0 = host payload, 1 = `data` is a device pointer on the handle's device (no PCIe copy); the single-cloud calls
 * (m3dreg_cloud_create, m3dreg_cloud_create_pc2) also take it as a flag word: | M3DREG_CLOUD_SOURCE_ONLY = the cloud will only ever be
 * a source (see m3dreg_cloud_desc.source_only). */
#define M3DREG_CLOUD_DEVICE 1
#define M3DREG_CLOUD_SOURCE_ONLY 2
int m3dreg_cloud_create(m3dreg_handle* h, const void* data, size_t n, size_t point_step,
                        size_t off_x, size_t off_y, size_t off_z, int data_is_device,
                        m3dreg_cloud** out);
/* Buckets many clouds at once: one decode launch and one bucketing pipeline serve the whole batch
 * (a single 100k-point cloud cannot fill 256 CUs). The grid geometry (from the exact AABB), the hash-table geometry and the
 * error state of every cloud are derived ON THE DEVICE and stay there, next to the cloud.
 *   m3dreg_cloud_create_batch        waits for the pipeline and reports M3DREG_ERR_EMPTY_CLOUD / M3DREG_ERR_GRID_TOO_LARGE like
 *                                    the single-cloud calls (no cloud is returned on failure);
 *   m3dreg_cloud_create_batch_async  enqueues only — NO host synchronisation: a registration on the same handle follows in
 *                                    stream order, one on another handle is ordered behind it on the device (an event per
 *                                    batch). Host payloads must stay valid until the pipeline has consumed them
 *                                    (m3dreg_synchronize, m3dreg_cloud_status, or the wait of a registration that uses the
 *                                    cloud). A cloud that turns out empty or too large ends every registration that names it
 *                                    with status M3DREG_BAD_CLOUD; m3dreg_cloud_status (waits) returns its error code, and so
 *                                    do m3dreg_cloud_grid_info / _export.
 * HBM held by a bucketed cloud of n points, per level (one pooled block per cloud, returned to the handle's pool on destroy): 72 B per point
 * (sorted points, keys, sorted keys, permutation, chunk boxes, block order, normals in sorted order when point-to-plane) + 32 B x the next
 * power of two >= 2n (bucket table; its used part is sized on the device) — plus, on the FINEST level of a cloud that can be a target,
 * 1.5 tile images of 56 KB per 512 points and a 1 MiB occupancy bitmap (164 B per point: 17 MB for a 100 k-point sweep, 250 MB for a
 * 1.5 M-point map) — plus 16 B (+ 16 B normals) per point in input order, shared by the levels. A source_only cloud builds its finest
 * level only, without table, tiles or normals. */
typedef struct m3dreg_cloud_desc {
    const void* data;     /* PointCloud2 payload (host, or device when data_is_device != 0) */
    size_t n, point_step, off_x, off_y, off_z;
    int32_t data_is_device;
    int32_t source_only;  /* != 0: the cloud will only ever be the SOURCE of registrations: it is sorted along the grid's curve (its queries
                             then stream spatially coherent) but gets no bucket table, no chunk boxes and no normals — the
                             normal-estimation grid alone is half of a point-to-plane cloud's bucketing. As a target it is refused
                             (M3DREG_ERR_LEVEL_MISMATCH); m3dmap_insert and the export calls take it. */
} m3dreg_cloud_desc;
int m3dreg_cloud_create_batch(m3dreg_handle* h, const m3dreg_cloud_desc* descs, size_t n_clouds, m3dreg_cloud** out);
int m3dreg_cloud_create_batch_async(m3dreg_handle* h, const m3dreg_cloud_desc* descs, size_t n_clouds, m3dreg_cloud** out);
int m3dreg_cloud_status(m3dreg_handle* h, const m3dreg_cloud* c);   /* M3DREG_OK or the error the device found; waits for the bucketing */
/* The whole sensor_msgs/PointCloud2 layout contract, decoded on the device: x / y / z are found by
 * NAME in the field table, the way pcl::fromPCLPointCloud2 resolves them in the consumer idiom of
 * m3d_aggregator.cpp:243-246; any offsets (aligned or not), FLOAT32 or FLOAT64 (rounded to nearest float; PCL itself
 * refuses to map a FLOAT64 x onto pcl::PointXYZ), either byte order, organised clouds with padded rows
 * (point i at (i / width) * row_step + (i % width) * point_step). Other fields (intensity, ring, rgb) are skipped.
 * M3DREG_ERR_INVALID_ARG names what is wrong through m3dreg_last_error (missing field, unsupported datatype, sizes that do
 * not fit data_bytes). The aggregator's own layout (16 / 0 / 4 / 8, little-endian) takes the coalesced fast path. */
#define M3DREG_FLOAT32 7   /* sensor_msgs/PointField.FLOAT32 */
#define M3DREG_FLOAT64 8   /* sensor_msgs/PointField.FLOAT64 */
typedef struct m3dreg_point_field { const char* name; uint32_t offset; uint8_t datatype; uint32_t count; } m3dreg_point_field;
int m3dreg_cloud_create_pc2(m3dreg_handle* h, const void* data, size_t data_bytes, uint32_t width, uint32_t height, uint32_t point_step,
                            uint32_t row_step, const m3dreg_point_field* fields, size_t n_fields, int is_bigendian, int data_is_device,
                            m3dreg_cloud** out);
int m3dreg_cloud_destroy(m3dreg_handle* h, m3dreg_cloud* c);
int m3dreg_align_clouds(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target,
                        const float init_T[16], float out_T[16], m3dreg_stats* stats);
/* Registers n_pairs independent pairs on this handle's device and waits for them: one launch chain, one launch per stage and Gauss-Newton iteration for the
 * whole batch. out_T: 16 * n_pairs floats. To keep the GPU busy ACROSS calls use the asynchronous pair below with several handles (bench.py's headline: the
 * bucketing of the next batch runs under the iterations of this one). Convergence-terminated batches (eps_rot / eps_trans > 0): this call — it waits for the
 * batch anyway — keeps its enqueue four iterations ahead of the device and stops a level as soon as the device reports it finished;
 * m3dreg_align_batch_async, which must not block, enqueues on and lets the launches behind a finished level leave at once (~5 us each). Same results. */
int m3dreg_align_batch(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs, float* out_T,
                       m3dreg_stats* stats);
/* (ABI 7) A caller that makes ONE call at a time on this GPU — the ROS node: one spin thread, one registration or one batch per sweep
 * (m3d_aggregator.cpp:185) — says so: on != 0 sizes this handle's launch grids for a GPU it has to itself (measurements: DESIGN.md §5). Leave it off when
 * other handles' batches share the GPU. Same results either way (the sums are integers); the library never guesses this from what the process has in
 * flight. Refused (M3DREG_ERR_INVALID_ARG) between m3dreg_align_batch_async and its wait. Default off. */
int m3dreg_set_latency_mode(m3dreg_handle* h, int on);
/* Enqueue only (no host sync); results are fetched by m3dreg_batch_wait. A handle holds the state of ONE batch: a second
 * m3dreg_align_batch_async before the wait is refused (M3DREG_ERR_INVALID_ARG) — to queue batches behind each other, give
 * several handles the same stream (m3dreg_create's `stream`), as bench.py does. */
int m3dreg_align_batch_async(m3dreg_handle* h, const m3dreg_pair* pairs, size_t n_pairs);
int m3dreg_batch_wait(m3dreg_handle* h, float* out_T, m3dreg_stats* stats);
int m3dreg_synchronize(m3dreg_handle* h);
void* m3dreg_get_stream(m3dreg_handle* h);

/* ---- one process, several GPUs (SURVEY.md §8 rows b / e) ------------------------------------------------------------
 * The consumer of this library is ONE process (m3d_husky_bringup.launch:13 starts one gpu_6dslam_node): a loop-closure batch is
 * spread over the devices it names, without torchrun and without a collective — the pairs are independent. m3dreg_multi_align
 * takes the raw payloads (a cloud lives on ONE device, so the sharding must come before the upload), assigns the pairs to the
 * devices longest-processing-time-first by their point counts (n_source + n_target — a shared target (target_group) counted once, its
 * pairs kept together —, at most ceil(n_pairs / n_devices) pairs per device where the groups allow it), uploads and buckets every shard on its own device and stream (sources source-only), enqueues all registrations, and
 * only then waits: the devices run concurrently, the results are gathered into the caller's arrays in PAIR order (a few hundred
 * bytes per pair over PCIe: no RCCL inside one process). `devices` may name a device more than once (several streams on it).
 * Results are bit-identical to m3dreg_align_batch on any one of the devices. */
typedef struct m3dreg_multi m3dreg_multi;
typedef struct m3dreg_pair_desc {
    m3dreg_cloud_desc source, target;   /* source.source_only is implied */
    float init_T[16];
    int32_t target_group;               /* (ABI 5) 0 = this pair's target is its own; > 0: every pair of the call with this id registers against the SAME
                                           reference cloud (identical target descriptors: loop-closure candidates against one submap): the
                                           group is kept on one device and its target is uploaded and bucketed ONCE there */
    int32_t reserved;                   /* must be 0 (ABI 6: checked). Zero-initialise the struct: m3dreg_pair_desc p = {0}; */
} m3dreg_pair_desc;
int m3dreg_multi_create(const m3dreg_params* params, const int* devices, int n_devices, m3dreg_multi** out);
int m3dreg_multi_destroy(m3dreg_multi* m);
int m3dreg_multi_align(m3dreg_multi* m, const m3dreg_pair_desc* pairs, size_t n_pairs, float* out_T /* 16 * n_pairs */,
                       m3dreg_stats* stats /* n_pairs, may be NULL */, int32_t* device_of_pair /* n_pairs, may be NULL: where each pair ran */);
const char* m3dreg_multi_last_error(const m3dreg_multi* m);
int m3dreg_debug_multi_clouds(const m3dreg_multi* m);   /* tests: clouds uploaded + bucketed by the last m3dreg_multi_align (a target group's target counts once) */
/* (ABI 4) Inside, every listed device has its own host thread, which uploads, buckets, registers and collects its shard, so the
 * devices' uploads and enqueues run side by side (one host thread + one HIP stream per device). The call itself is
 * synchronous and may be made from any ONE thread at a time per context. Host payloads cross PCIe from pinned memory: a payload
 * that lives in memory from m3dreg_host_alloc (or registered with m3dreg_host_register, or pinned by any other HIP call of the
 * process) is copied by the DMA engine as it is, asynchronously; a pageable one is first copied into the device thread's pinned
 * staging block. No exception and no half-enqueued state survive an error: whatever a device thread enqueued is waited for, its
 * clouds are released and its handle is idle again before the call returns (M3DREG_ERR_OUT_OF_MEMORY: the context stays usable). */

/* ---- pinned host memory without linking HIP ---------------------------------------------------------------------------------
 * A PointCloud2 payload handed over from pageable memory is copied synchronously and staged by the runtime; from pinned memory the
 * copy is asynchronous (it overlaps the previous batch's kernels and the other devices' copies). m3dreg_host_alloc / _free wrap
 * hipHostMalloc / hipHostFree (portable: valid for every device), m3dreg_host_register / _unregister pin an existing range in place
 * (the ROS shim's message buffers, a bag reader's pool). Every entry point that takes a host payload accepts either kind. */
int m3dreg_host_alloc(size_t bytes, void** out);
int m3dreg_host_free(void* p);
int m3dreg_host_register(void* p, size_t bytes);
int m3dreg_host_unregister(void* p);

/* ---- aggregation on the device ---------------------------------------------
 * The step m3d_aggregator performs before publishing a cloud (m3d/m3d_aggregator/src/m3d_aggregator.cpp):
 * every incoming PointCloud2 / LaserScan message is rigidly transformed by the tf lookup of its callback
 * (:236-248, :261-268), points INSIDE the self-filter box are dropped (:65-73), the rest is appended, and the
 * rotation travelled by the head is accumulated (:75-87) until it exceeds 1.1*pi (:30, :95-103).
 * `tf7` = {tx, ty, tz, qx, qy, qz, qw} of geometry_msgs/Transform. The aggregate lives in HBM in
 * pcl::PointXYZ layout, so m3dagg_take_cloud buckets it in place — the sweep never crosses PCIe as a cloud. */
typedef struct m3dagg m3dagg;
/* bbox = {x_up, x_down, y_up, y_down, z_up, z_down} (setBBox :42-52; node defaults +-1 m, :164-171) */
int m3dagg_create(m3dreg_handle* h, const double bbox[6], size_t capacity, m3dagg** out);
int m3dagg_destroy(m3dagg* a);
/* rotLaserPointCloudCallback (:231-254): `data` is the message's host buffer */
int m3dagg_add_cloud(m3dagg* a, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                     const double tf7[7]);
/* rotLaserScanCallback (:256-288): ranges[i] at angle_min + i * angle_increment, z = 0 */
int m3dagg_add_scan(m3dagg* a, const float* ranges, size_t n, float angle_min, float angle_increment, const double tf7[7]);
/* (ABI 5) which function `point.x = cos(ang)*dist` (:281-282, float operands, unqualified call) resolves to: 0 (default) = C's
 * double cos(double) — ang promoted, the product formed in double, rounded once into the float field: what GCC < 6 / the ROS1-era
 * toolchains do; 1 = the float overload (cosf(ang) * dist in float: GCC >= 6 with the C++ <math.h> wrapper visible). */
int m3dagg_set_scan_trig(m3dagg* a, int float_overload);
/* getProgress (:119-124) in percent, isPointcloudReady (:95-103), currentAngularDistance, points kept so far */
int m3dagg_status(m3dagg* a, double* progress, int* ready, double* angle, size_t* n_points);
/* publishPointcloud (:194-212) without the publish: buckets the aggregate as an m3dreg_cloud, then clears and
 * restarts the aggregator (clearPointCloud :108-114; the node is re-armed by requestCallback :224-229). */
int m3dagg_take_cloud(m3dagg* a, m3dreg_cloud** out);
int m3dagg_restart(m3dagg* a);                       /* requestCallback (:224-229) */
/* (ABI 8) automatic != 0 (default): m3dagg_take_cloud re-arms the aggregator itself; 0: it leaves the aggregator idle — messages are ignored (:55), progress
 * reads -1 (:121) — until m3dagg_restart, exactly like the reference's node between a published cloud and the next ~request (:211, :224-229). */
int m3dagg_set_rearm(m3dagg* a, int automatic);
int m3dagg_download(m3dagg* a, float* xyzw, size_t cap_points, size_t* n_out);   /* tests: 16 bytes per point */

/* ---- calibration cost on the device ------------------------------------------
 * The cost function the reference's two calibration nodes minimise (`testData`,
 * m3d/m3d_calibration/src/m3d_calibration_twiddle.cpp:199-308 = m3d_calibration_sa.cpp:199-277): every scan
 * segment of a calibration sweep (scanSegment: points in the laser frame + the tf of its message, :33-38, :56-69) is
 * moved by original_Transform * laserOffsetMatrix (:229-230), the points are split on the sign of their RAW
 * coordinate along `laser_up_axis` (:234-266), both halves are voxel-grid filtered at 0.1 m (:279-286) and the
 * cost is the number of second-half voxel centroids with no first-half centroid within 0.05 m (:288-304).
 * The library evaluates many candidates per launch; the two optimiser loops are host code with the reference's
 * control flow and constants. */
typedef struct m3dcal m3dcal;
int m3dcal_create(m3dreg_handle* h, int laser_up_axis, m3dcal** out);       /* laserUpAxis param (:176), 0 / 1 / 2 */
int m3dcal_destroy(m3dcal* c);
/* addPoints (:56-69): one scan segment = n points (FLOAT32 x/y/z inside point_step, host buffer) + its
 * original_Transform as column-major float[16] (Eigen::Affine3f::data()) */
int m3dcal_add_segment(m3dcal* c, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                       const float original_T[16]);
/* testData for k candidates at once: params = k x {x, y, z, yaw, pitch, roll}; counts[k] receives `c`;
 * voxels (optional) k x {voxels of firstPcFilter, voxels of secondPcFilter} */
int m3dcal_evaluate(m3dcal* c, const float* params, size_t k, int64_t* counts, int64_t* voxels);
/* publishPointcloud of the twiddle node (:330-396): p = 0, dp = 0.01, x1.1 on improvement, x0.9 on failure, stops when
 * the LAST dp falls to 1e-6 (:376-380, :348) or after max_sweeps (0 = no limit). p_out[5] = (y, z, yaw, pitch, roll) as
 * the node maps them (:345: testData(0, p0, p1, p2, p3, p4)); returns the number of sweeps through *sweeps. */
int m3dcal_twiddle(m3dcal* c, int max_sweeps, float p_out[5], float* best_error, int* sweeps, int* evaluations);
/* publishPointcloud of the annealing node (m3d_calibration_sa.cpp:284-356): T0 = 1, alpha = 0.99, until T <= 0.001
 * (688 evaluations), proposal p + 0.001 * U(-1, 1) per coordinate, Metropolis acceptance, rand()/RAND_MAX after
 * srand(seed) (the reference seeds with time(0), :289). p_io: start on entry ((0, 0.12, 0, 0, 0) in the node, :303-308). */
int m3dcal_anneal(m3dcal* c, unsigned int seed, float p_io[5], float* best_error, int* evaluations);

/* ---- persistent map in HBM ---------------------------------------------------
 * The dense aggregated reference of BASELINE config 5, maintained on the device: every registered scan is inserted with
 * its pose, only points whose dedup voxel (floor(u / dedup_leaf)) is still empty are kept (lowest input index of the scan
 * wins, kept points are appended in input order — deterministic), and m3dmap_as_cloud buckets the point buffer in place
 * as a registration target. Nothing of the map ever crosses PCIe. No reference source exists for this step (the
 * reference's gpu_6dslam is an empty submodule); the behaviour is specified in DESIGN.md §8 and restated by
 * oracle/m3d_map_oracle.c. */
typedef struct m3dmap m3dmap;
int m3dmap_create(m3dreg_handle* h, float dedup_leaf, size_t capacity_points, m3dmap** out);
int m3dmap_destroy(m3dmap* m);
/* T: pose of the scan in the map frame, column-major float[16] (what m3dreg_align returned, chained) */
int m3dmap_insert(m3dmap* m, const m3dreg_cloud* scan, const float T[16], size_t* n_added);
int m3dmap_size(m3dmap* m, size_t* n_points);
int m3dmap_as_cloud(m3dmap* m, m3dreg_cloud** out);          /* the map as a bucketed target (the map itself keeps growing) */
int m3dmap_download(m3dmap* m, float* xyzw, size_t cap_points, size_t* n_out);   /* tests: 16 bytes per point */
int m3dmap_clear(m3dmap* m);

/* ---- loop-closure candidate generation (second half; ABI 8) -----------------
 * Decides WHICH pairs a loop-closure batch registers — the step in front of m3dreg_align_batch / m3dreg_multi_align (BASELINE config 4:
 * "batch of 64 loop-closure scan pairs"). No reference source exists (the node the reference launches is called gpu_6dslam_node,
 * m3d_husky_bringup.launch:13, and its repository is an empty submodule): the behaviour is this library's own, specified in DESIGN.md §10
 * and restated by oracle/m3d_loop_oracle.c.
 *   keyframe k  = (pose T_k of the sweep in the map frame, a bucketed cloud resident in HBM), numbered in insertion order;
 *   signature   = a bitmap of 2^sig_log2_bits bits: every finite point p of the cloud sets the bit hash(floor((R_k p + t_k) / sig_leaf)) —
 *                 the coarse voxels of the MAP frame the sweep saw (built on the device from the cloud's resident points, never downloaded);
 *   candidates  of keyframe i: every older keyframe j with i - j >= min_gap and |t_i - t_j| <= radius whose overlap
 *                 popcount(sig_i & sig_j) is at least min_overlap * min(popcount(sig_i), popcount(sig_j)); the top_k of them by overlap (ties: the
 *                 older keyframe first). All integer arithmetic: the result does not depend on launch geometry.
 * A candidate names source = i (the later sweep), target = j, and carries init_T = inv(T_j) * T_i, the odometry's guess of source -> target.
 * m3dloop_make_pairs turns candidates into m3dreg_pair[] (the keyframes' resident clouds) for m3dreg_align_batch;
 * m3dloop_make_pair_descs into m3dreg_pair_desc[] (the payload descriptors given with the keyframes, target_group = target + 1: pairs against
 * one keyframe stay on one device and its cloud is bucketed once) for m3dreg_multi_align. m3dloop_gate is the acceptance test on the
 * registration's statistics. Keyframe clouds stay the caller's: they must outlive the m3dloop (or its m3dloop_clear) and must not be
 * source-only when they are to be targets. */
typedef struct m3dloop m3dloop;
typedef struct m3dloop_params {
    float sig_leaf;          /* edge of the signature's voxels [m] (default 2.0: coarse enough to survive the drift a loop accumulates) */
    int32_t sig_log2_bits;   /* signature size = 2^this bits, 10 .. 18 (default 16: 8 KB per keyframe) */
    float radius;            /* [m] keyframe positions at most this far apart (default 10) */
    int32_t min_gap;         /* >= 1: keyframes at least this many insertions apart (default 10: not the odometry's own neighbours) */
    int32_t top_k;           /* 1 .. 16 candidates per keyframe at most (default 2) */
    float min_overlap;       /* 0 .. 1 (default 0.5) */
    int32_t max_keyframes;   /* capacity, 1 .. 65536 (default 4096: 32 MB of signatures) */
    int32_t reserved;        /* must be 0 */
} m3dloop_params;
typedef struct m3dloop_candidate {
    int32_t source, target;            /* keyframe indices, source > target */
    uint32_t overlap;                  /* popcount(sig_source & sig_target) */
    uint32_t pop_source, pop_target;   /* popcount of either signature */
    float dist2;                       /* |t_source - t_target|^2 as the prefilter computed it */
    float init_T[16];                  /* column-major inv(T_target) * T_source */
} m3dloop_candidate;
int m3dloop_default_params(m3dloop_params* out);
int m3dloop_create(m3dreg_handle* h, const m3dloop_params* params, m3dloop** out);
int m3dloop_destroy(m3dloop* l);
int m3dloop_clear(m3dloop* l);
/* `payload` (may be NULL): the descriptor of the sweep's raw PointCloud2 payload, kept by value for m3dloop_make_pair_descs. */
int m3dloop_add_keyframe(m3dloop* l, const m3dreg_cloud* cloud, const float T[16], const m3dreg_cloud_desc* payload, int32_t* index);
/* a pose graph moved keyframe `index`: its signature is rebuilt from its cloud at the new pose */
int m3dloop_update_pose(m3dloop* l, int32_t index, const float T[16]);
int m3dloop_size(m3dloop* l, size_t* n_keyframes);
/* Candidates of keyframes first .. first + count - 1 (count < 0: to the newest), row by row, best first inside a row; at most `cap` are
 * written, *n_out receives how many exist. A node calls it with the keyframe it has just added (one row against the whole database: one
 * streaming pass over the older signatures); a back end that rebuilds its graph calls it for all rows. Waits for the device. */
int m3dloop_candidates(m3dloop* l, int32_t first, int32_t count, m3dloop_candidate* out, size_t cap, size_t* n_out);
int m3dloop_make_pairs(m3dloop* l, const m3dloop_candidate* cands, size_t n, m3dreg_pair* out);
int m3dloop_make_pair_descs(m3dloop* l, const m3dloop_candidate* cands, size_t n, m3dreg_pair_desc* out);
/* accept[i] = 1 iff the registration ended CONVERGED or MAX_ITERATIONS with n_corr >= min_corr and rms <= max_rms */
int m3dloop_gate(const m3dloop_candidate* cands, const m3dreg_stats* stats, size_t n, int64_t min_corr, double max_rms, uint8_t* accept);
/* tests: the signature words (2^sig_log2_bits / 32 of them) and its popcount */
int m3dloop_signature(m3dloop* l, int32_t index, uint32_t* words, uint32_t* pop);
/* device time of the last m3dloop_candidates call's kernels [ms] (hipEvents on the handle's stream) and the bytes of signatures they had to read */
int m3dloop_last_profile(m3dloop* l, double* ms, uint64_t* algorithmic_bytes);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Per-stage device times of the path ("ms per stage"). on = 0: off; on = n >= 1: every n-th Gauss-Newton iteration
 * of the handle is bracketed by three hipEvents on its stream: before and after the correspondence step (`k_nn_iter`: certificate
 * check, binning / sparse search of every query of every pair of the batch; `k_nn_tiles`: the binned searches from LDS) and at
 * the end of the iteration (reduction + solve), and every bucketing batch by two, so bench.py can report the stage durations
 * from inside its timed region. (An event record is a barrier packet on the queue: bracketing every iteration cost 4 % of the
 * throughput it measured.) With M3DREG_ROCTX=1 in the environment the same stages are also marked as roctx ranges
 * ("m3dreg:bucketing", "m3dreg:iteration") for rocprofv3 --marker-trace. */
int m3dreg_profile_enable(m3dreg_handle* h, int on);
/* (ABI 8) The per-BATCH brackets (two events around a bucketing batch, two around a batch's whole chain of iterations) on every n-th batch of the handle only
 * (default 1: every batch; m3dreg_profile_enable does not change it). An event record is a barrier packet on the queue: with 4 per step + the iteration brackets
 * of every 7th iteration the headline measured 5 % less than without any (profiles/r06_event_density.txt); bench.py samples every 4th batch and every 13th iteration. */
int m3dreg_profile_batches(m3dreg_handle* h, int every);
#define M3DREG_PROFILE_ITERATION 0        /* one Gauss-Newton iteration of the batch: correspondence step + reduction + solve */
#define M3DREG_PROFILE_DOMINANT_KERNEL 1  /* its correspondence step (a6) */
#define M3DREG_PROFILE_BUCKETING 2        /* one bucketing batch (a2-a4, a9, tiles): m3dreg_cloud_create* */
#define M3DREG_PROFILE_REDUCE_SOLVE 3     /* a7 + a8 of the bracketed iterations (= ITERATION - DOMINANT_KERNEL) */
#define M3DREG_PROFILE_CHAIN 4            /* ALL iterations of a batch as they ship (fused late launches included), one bracket per batch:
                                             n_launches counts the iterations enqueued, total_ms / n_launches = mean time of a whole linearisation */
/* Synchronises the stream, returns the number of bracketed launches of kind `what` and the sum of their
 * durations since the last reset; `reset` != 0 clears that kind's counters. */
int m3dreg_profile_read(m3dreg_handle* h, int what, uint64_t* n_launches, double* total_ms, int reset);

/* ---- introspection used by the parity tests (stage-by-stage comparison with oracle/) -------- */
typedef struct m3dreg_grid_info {
    int32_t n;          /* points in the input cloud */
    int32_t n_valid;    /* finite points */
    int32_t n_cells;    /* occupied voxels */
    int32_t dims[3];    /* grid extent in cells */
    int32_t bits[3];    /* key bit-field widths: key = ix | iy<<bits[0] | iz<<(bits[0]+bits[1]) */
    float mn[3];        /* AABB min (grid origin) */
    float mx[3];        /* AABB max */
    float center[3];    /* linearisation centre c */
    float leaf;
    float inv_leaf;
    float lbound;       /* bound on |w| components used to derive the fixed-point scales */
    int32_t has_normals;
} m3dreg_grid_info;

int m3dreg_cloud_levels(const m3dreg_cloud* c);
int m3dreg_cloud_grid_info(m3dreg_handle* h, const m3dreg_cloud* c, int level, m3dreg_grid_info* out);
/* Any output pointer may be NULL. Arrays are sized by the cloud's n:
 *   keys[n]       voxel key per input point, input order (0xFFFFFFFF for non-finite points)
 *   sorted_keys[n], perm[n]   stable sort by key; perm[j] = input index of the j-th sorted point
 *   sorted_xyz[3n]            x,y,z of sorted points (interleaved)
 *   normals[3n]               unit normal per sorted point, (0,0,0) = invalid; only if has_normals */
/* (ABI 5) mean population of the voxel a point of the cloud lies in, at `level` (sum over the occupied voxels of population^2 / finite points; derived by
 * the bucketing pipeline on the device): what a query meets in its home voxel. Registration time on an MI355X follows it (0.72 ... 1.40 ms over the 64 pairs
 * of BASELINE config 4, correlation 0.9): cost(pair) = density(source) + density(target) is the a-priori estimate a caller shards a batch by
 * (mandala_mapping_amd/sharding.py lpt_assign; bench.py --gpus N). Waits for the cloud's bucketing. */
int m3dreg_cloud_density(m3dreg_handle* h, const m3dreg_cloud* c, int level, double* out);
int m3dreg_cloud_export(m3dreg_handle* h, const m3dreg_cloud* c, int level, uint32_t* keys,
                        uint32_t* sorted_keys, int32_t* perm, float* sorted_xyz, float* normals);
/* NN of arbitrary queries (already in the target frame; host float xyz interleaved) against one
 * level of a bucketed cloud: out_idx[i] = INPUT index of the match or -1, out_d2[i] = squared distance. */
int m3dreg_debug_nn(m3dreg_handle* h, const m3dreg_cloud* target, int level, const float* queries_xyz,
                    size_t nq, float max_corr_dist, int32_t* out_idx, float* out_d2);
/* How many candidates the spec names for every query: the points of the 27 voxels around it (what an exhaustive search compares;
 * the product's searches prune most of them). bench.py's gather-model bytes use its mean. */
int m3dreg_debug_candidates(m3dreg_handle* h, const m3dreg_cloud* target, int level, const float* queries_xyz,
                            size_t nq, int32_t* out_count);
/* One linearisation at pose T (no update): the 29 fixed-point sums and their power-of-two exponents
 * (value = sum * 2^-exp; exps[6] = rr, rt, tt, gr, gt, ss). */
int m3dreg_debug_accumulate(m3dreg_handle* h, const m3dreg_cloud* source, const m3dreg_cloud* target,
                            int level, const float T[16], int64_t sums[M3DREG_NSUMS], int32_t exps[6]);
/* Per-iteration pose trace of the most recent m3dreg_align/align_clouds on this handle:
 * column-major double[16] after each executed iteration; returns count via *n_out (<= cap). */
int m3dreg_debug_trace(m3dreg_handle* h, double* poses, size_t cap, size_t* n_out);
/* Diagnostics of the first pair of the most recent batch: out[0] = searches answered from LDS-staged tiles (k_nn_tiles),
 * out[1] = searches of flagged tiles / full slabs walked in global memory instead. */
int m3dreg_debug_counters(m3dreg_handle* h, uint64_t out[2]);
/* Test hooks of the exception guard: the n-th host allocation of the library from now on throws std::bad_alloc (0 = off);
 * m3dreg_debug_throw runs a guarded body that throws (kind 0: std::bad_alloc, else: something else) and returns what the
 * boundary made of it (M3DREG_ERR_OUT_OF_MEMORY / M3DREG_ERR_HIP). Neither needs a device. */
int m3dreg_debug_fail_alloc(int nth);
/* (ABI 8) Report of the DIAGNOSIS build (csrc: `make checked` -> libm3dreg_checked.so, -DM3D_CHECKED): every index a kernel reads from memory is compared with its
 * bound before it addresses global memory; out[0..3] = {offences, site, index, bound} of the iteration kernels, out[4..7] of the bucketing pipeline (first offence
 * kept; reset != 0 clears). The shipped library has no checks compiled in: M3DREG_ERR_INVALID_ARG. */
int m3dreg_debug_checks(m3dreg_handle* h, uint32_t out[8], int reset);
/* (ABI 8) Diagnosis: the raw bytes of one of a cloud level's search structures as they lie in HBM (waits for the cloud's bucketing) — what = 0 bucket table
 * (32 B per slot), 1 tile headers (16 B per tile), 2 tile images, 3 tile image meta (8 B per image), 4 occupancy bitmap, 5 the level's meta (dyn counters + grid),
 * 6 the source block order. *bytes = size of the structure; copied when out != NULL and cap >= *bytes. The layouts are csrc/m3d_device.h's, not part of the ABI:
 * scripts/r6_hunt2.py compares two clouds bucketed from the same points with it. M3DREG_ERR_LEVEL_MISMATCH when the cloud has no such structure. */
int m3dreg_debug_cloud_raw(m3dreg_handle* h, const m3dreg_cloud* c, int level, int what, void* out, size_t cap, size_t* bytes);
int m3dreg_debug_throw(int kind);

#ifdef __cplusplus
}
#endif
#endif /* M3DREG_H */
