/*
 * fdtd_abi.h — C ABI of the MI355X (gfx950) 3D acoustic FDTD engine.
 *
 * Two layers are declared here, both plain C (pointers and sizes only):
 *
 *   1. The Devito-style operator ABI the reference driver binds
 *      (reference main.cpp:35-50 for the structs, main.cpp:53-72 for the three
 *      24-argument operators).  A program written against the reference links
 *      against libfdtd_hip.so without source changes.
 *
 *   2. A device-resident ABI (fdtd_hip_*): the same two device kernels the
 *      reference launches per time step (cuda.cu:61-108 and cuda.cu:112-170 /
 *      cuda_optimized.cu:63-260), callable on fields that already live in HBM.
 *      The operators of layer 1 are built on it; callers that keep the wavefield on
 *      the GPU between steps (the benchmark, the multi-GPU slab propagator) use it
 *      directly.
 *
 * Memory layout (identical on host and device, reference cuda.cu:38-42):
 *   u[t][x][y][z]  float32, t in 0..2, z contiguous; every spatial axis is padded
 *   by FDTD_HALO cells on both sides, i.e. interior cell i lives at padded index
 *   i + FDTD_HALO (openacc.cpp:102).  m[x][y][z] has the same padded shape.
 *   src[time][p], src_coords[p][3].
 */
#ifndef FDTD_ABI_H
#define FDTD_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Space order of the caller's arrays: 4 (radius-2 star stencil, padding of 4 cells per side) unless the
 * including program is compiled with -DFDTD_STENCIL_ORDER=6, 8, 10 or 12, the way the reference driver takes
 * -DSTENCIL_ORDER with HALO = STENCIL_ORDER (main.cpp:27-32; cuda.cu:55).  The macros describe the CALLER's
 * arrays; the library itself takes the order per call or per process (fdtd_hip_set_space_order). */
#ifndef FDTD_STENCIL_ORDER
#define FDTD_STENCIL_ORDER 4
#endif
#define FDTD_RADIUS (FDTD_STENCIL_ORDER / 2)
#define FDTD_HALO FDTD_STENCIL_ORDER
/* The first FDTD_WARMUP_STEPS time steps of a call are real steps that are not
 * timed (openacc.cpp:5,90-93; cuda.cu:56,232). */
#define FDTD_WARMUP_STEPS 5

/* main.cpp:35-45.  Only data, size and nbytes are ever populated by the driver
 * (main.cpp:114-126, 540-545); the remaining fields are carried for layout
 * compatibility (72 bytes on LP64). */
struct dataobj {
  void *data;
  int *size;
  unsigned long nbytes;
  unsigned long *npsize;
  unsigned long *dsize;
  int *hsize;
  int *hofs;
  int *oofs;
  void *dmap;
};

/* main.cpp:47-50.  Seconds spent in the stencil (section0) and in the source
 * injection (section1) over time steps time_m+FDTD_WARMUP_STEPS .. time_M. */
struct profiler {
  double section0;
  double section1;
};

/* ---------------------------------------------------------------------------
 * Layer 1: operator ABI (main.cpp:53-72).  Max comes before min in every pair;
 * all ranges are inclusive.  Host buffers in, host buffers out: u_vec->data is
 * updated in place (all three time levels, halos included), m/src/src_coords are
 * read-only.  Returns 0 on success, a positive hipError_t value on failure
 * (cuda.cu:280-284; hipErrorInvalidValue for inconsistent shapes,
 * cuda_optimized.cu:346-347).  Timers are accumulated (+=) like openacc.cpp:3 and
 * cuda.cu:11-16, so the caller zeroes them (main.cpp:390).
 *
 *   Kernel_OpenACC        replaces openacc.cpp:61-216  (flat collapse(3) sweep, the
 *                         shape an OpenACC compiler gives the loop nest)
 *   Kernel_CUDA           replaces cuda.cu:173-323     (one thread per cell, global
 *                         memory only, z on the lanes)
 *   Kernel_CUDA_Optimized replaces cuda_optimized.cu:282-514 (LDS-staged x-marching
 *                         tile kernel; the default fast path)
 * ------------------------------------------------------------------------- */
#define FDTD_OPERATOR_ARGS                                                        \
  struct dataobj *m_vec, struct dataobj *src_vec, struct dataobj *src_coords_vec, \
      struct dataobj *u_vec, const int x_M, const int x_m, const int y_M,         \
      const int y_m, const int z_M, const int z_m, const float dt,                \
      const float h_x, const float h_y, const float h_z, const float o_x,         \
      const float o_y, const float o_z, const int p_src_M, const int p_src_m,     \
      const int time_M, const int time_m, const int deviceid, const int devicerm, \
      struct profiler *timers

int Kernel_OpenACC(FDTD_OPERATOR_ARGS);
int Kernel_CUDA(FDTD_OPERATOR_ARGS);
int Kernel_CUDA_Optimized(FDTD_OPERATOR_ARGS);

/* Same operator with the three wavefield levels stored as IEEE binary16 in HBM and
 * all arithmetic in fp32 (the reference's USE_FP32_ONLY=0 build,
 * cuda_optimized.cu:59-61,261-279,482-499).  Host buffers stay fp32. */
int Kernel_HIP_FP16Storage(FDTD_OPERATOR_ARGS);

/* The same operator on several GPUs of one node, in one process (one issuing host thread per slab while the
 * call lasts; see the fdtd_hip_multigpu_* session below, on which it is built): the x range is cut
 * into one slab per device, every pair of time steps ends with a push of 4 ghost planes to each
 * neighbour (hipMemcpyPeerAsync: copy engines over xGMI, no compute units), overlapped with the
 * interior sweep on a second stream.  No counterpart in the reference, which is single-device; the
 * result is bit for bit that of Kernel_CUDA_Optimized.  Devices: deviceid = -1 takes all visible
 * ones (FDTD_NGPUS limits their number; with deviceid >= 0 that many starting there, default one);
 * FDTD_MULTIGPU_DEVICES="0,1,2" names them (a device named twice carries two slabs).  A call whose x
 * range is not the whole extent of the arrays, or with fewer than 16 planes, runs on the first
 * device alone.  timers->section0 receives the wall time of steps time_m+5 .. time_M over all
 * devices (host clock between device-wide synchronisations), section1 nothing: the scatter is part
 * of the sweep. */
int Kernel_HIP_MultiGPU(FDTD_OPERATOR_ARGS);
/* Number of slabs the last Kernel_HIP_MultiGPU call was cut into (1: handed to one device). */
int fdtd_hip_multigpu_last_slab_count(void);

/* Device copies across calls.  openacc.cpp:208-212 copies u out and releases it at the end of every call and
 * deletes m, src and src_coords only `if(devicerm)`; the reference's CUDA operators ignore devicerm and upload
 * everything on every call (cuda.cu:204-214,321).  The same here: Kernel_OpenACC with devicerm = 0 leaves m (and
 * the device allocations) behind, and a later call with the same m pointer, padded shape and device computes on
 * that copy; u, src and src_coords are uploaded on every call, u is copied back at the end of every call.  The
 * other operators behave like that only with FDTD_RESIDENT_ALL_OPERATORS=1 in the environment; by default they
 * upload and free everything whatever devicerm says.  An OpenACC runtime would compute on stale data if the host
 * array was freed and allocated again at the same address or edited in place; here the entry also carries a
 * content tag of m (hash of its first and last 4 KB and of 4096 words spread over it), and a tag that differs
 * makes the call upload m again.  FDTD_RESIDENT_KEEP_U=1 (not reference behaviour) keeps u on the device too:
 * a later call with the same u pointer continues from the device copy without reading the host's u.
 * devicerm != 0 frees the copies at the end of the call; fdtd_hip_release_device_copies frees whatever earlier
 * calls left behind.  Kernel_HIP_Forward and Kernel_HIP_MultiGPU always upload and free everything. */
void fdtd_hip_release_device_copies(void);
/* Which host arrays the last operator call copied to the device: out = {u, m, src, src_coords}, 1 or 0 each. */
void fdtd_hip_last_call_uploads(int out[4]);
/* Host-side phase times of the last operator call, in milliseconds: allocation, copies to the
 * device, time loop, copy to the host. */
void fdtd_hip_last_call_phases(double out_ms[4]);

/* Space order of the operators and of fdtd_hip_time_loop: 4 (default), 6, 8, 10 or 12, the values the
 * reference driver's STENCIL_ORDER can take, with HALO = STENCIL_ORDER cells of padding per side
 * (main.cpp:27-32,129-136): a driver built with -DSTENCIL_ORDER=8 hands over arrays padded by 8 cells
 * and interior cell i lives at padded index i + 8.  No reference kernel implements any order but 4
 * (openacc.cpp:102-107), so the higher orders have no reference arithmetic (parity unpinned): they
 * evaluate the order-4 expression of openacc.cpp:159-164 with the central second-difference weights
 * of their order (9-digit float literals, farthest neighbours first) — boxes of 12 M cells and more
 * through a sweep that marches along x with the 2R+1 planes of the x direction in registers and the
 * current plane in LDS, smaller ones through a one-cell-per-thread kernel; same bits.  The order is
 * process-wide: set it before the operators are called (or export FDTD_STENCIL_ORDER); fp16 storage,
 * the two-step sweep and the multi-GPU operator exist for order 4 only.  Any other value is refused
 * (non-zero return) and leaves the order as it was. */
int fdtd_hip_set_space_order(int order);
int fdtd_hip_get_space_order(void);
/* The same with the order given per call, for programs that run operators of several orders, or from several
 * threads: Kernel_HIP_SpaceOrder is Kernel_CUDA_Optimized on arrays padded by space_order cells per side,
 * Kernel_HIP_ForwardOrder (declared with Kernel_HIP_Forward) and fdtd_hip_time_loop_order (with
 * fdtd_hip_time_loop) likewise.  The order holds for that call on the calling thread only; the process-wide
 * setting is neither read nor changed.  An unsupported order is refused with hipErrorInvalidValue. */
int Kernel_HIP_SpaceOrder(FDTD_OPERATOR_ARGS, int space_order);
/* The forward operator of a Devito-style acoustic propagator around the same path: the update with an
 * absorbing sponge (damp, padded shape of m), the source scatter, and receivers that sample the wavefield
 * (rec[time][p] at rec_coords[p][3], trilinear, the level the step starts from).  The reference has
 * neither sponge nor receivers (fixed halo frame, main.cpp:352; two sections only, openacc.cpp:148-205),
 * so there is no reference arithmetic for this entry: parity unpinned.  With damp = 0 the update is the
 * expression of openacc.cpp:159-164; the interpolation is the transpose of the scatter of
 * openacc.cpp:185-196.  Argument conventions of the operator ABI; p_rec_M < p_rec_m: no receivers.
 * Every space order (fdtd_hip_set_space_order).  At order 4, boxes the two-step sweep is made for (16 M cells
 * and more, rows of whole float4s) run two damped steps per sweep (fdtd_hip_two_steps_damped: 12 instead of 20
 * bytes per cell update); otherwise boxes of 12 M cells and more take the marching sweep of the space orders with
 * the damping field as one more streamed operand, smaller ones a one-cell-per-thread kernel: same bits. */
int Kernel_HIP_Forward(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                       struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                       struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M,
                       const int x_m, const int y_M, const int y_m, const int z_M, const int z_m,
                       const float dt, const float h_x, const float h_y, const float h_z,
                       const float o_x, const float o_y, const float o_z, const int p_rec_M,
                       const int p_rec_m, const int p_src_M, const int p_src_m, const int time_M,
                       const int time_m, const int deviceid, const int devicerm,
                       struct profiler *timers);

int Kernel_HIP_ForwardOrder(struct dataobj *damp_vec, struct dataobj *m_vec, struct dataobj *rec_vec,
                            struct dataobj *rec_coords_vec, struct dataobj *src_vec,
                            struct dataobj *src_coords_vec, struct dataobj *u_vec, const int x_M,
                            const int x_m, const int y_M, const int y_m, const int z_M, const int z_m,
                            const float dt, const float h_x, const float h_y, const float h_z,
                            const float o_x, const float o_y, const float o_z, const int p_rec_M,
                            const int p_rec_m, const int p_src_M, const int p_src_m, const int time_M,
                            const int time_m, const int deviceid, const int devicerm,
                            struct profiler *timers, int space_order);
/* 1 when the last Kernel_HIP_Forward call ran two steps per sweep, 0 when it ran one sweep per step. */
int fdtd_hip_forward_last_call_fused(void);

/* Optional hook the reference driver looks up as a weak symbol and calls once per
 * benchmark sweep (main.cpp:84,266-276).  The call records the values and returns.
 * t_fuse >= 2 makes Kernel_CUDA_Optimized take the two-steps-per-sweep kernel whatever
 * the x extent (at 1, the driver's default, the choice is automatic; the results are
 * the same bits).  use_tc and nfields select nothing: no tensor-core path exists for a
 * 13-point star, and the operators take one field. */
void FDTD_SetRuntimeConfig(int use_tc, int t_fuse, int nfields);

/* ---------------------------------------------------------------------------
 * Layer 2: device-resident ABI.  All pointers are device pointers; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 * with respect to the host.  Return value: 0 or a positive hipError_t.
 * ------------------------------------------------------------------------- */

/* Stencil variants selectable at run time. */
enum fdtd_variant {
  FDTD_VARIANT_FLAT = 0,     /* flat collapse(3) index, one cell per thread        */
  FDTD_VARIANT_NAIVE = 1,    /* 3-D grid, z on lanes, global loads only (cuda.cu)   */
  FDTD_VARIANT_MARCH = 2,    /* x-marching, register column + LDS centre plane      */
  FDTD_VARIANT_TILE3D = 3,   /* 8x8x8 outputs from a 12x12x12 LDS tile              */
  FDTD_VARIANT_VEC = 4,      /* one-shot float4 lanes, cache-ordered launch          */
  FDTD_VARIANT_MARCH2 = 5,   /* two time steps per sweep (fdtd_hip_two_steps and the time loop only) */
  FDTD_VARIANT_ROWS = 6,     /* one-shot float4 lanes, rows packed into waves (small and medium grids) */
  FDTD_VARIANT_AUTO = -1     /* MARCH from 16 M cells on when the shape allows it, else ROWS, else NAIVE */
};

/* Geometry of one padded field set; passed by pointer. */
struct fdtd_geom {
  int nxp, nyp, nzp;          /* padded extents (u_vec->size[1..3])                 */
  int x_m, x_M, y_m, y_M, z_m, z_M; /* inclusive interior ranges to update         */
  float dt, h_x, h_y, h_z;    /* time step and grid spacings                        */
  float o_x, o_y, o_z;        /* grid origin                                        */
};

/* One sweep / one scatter of a given space order on arrays padded by `order` cells per side. */
int fdtd_hip_stencil_step_order(const struct fdtd_geom *g, const float *m, float *u, int t0, int t1,
                                int t2, int order, void *stream);
int fdtd_hip_source_inject_order(const struct fdtd_geom *g, const float *m, const float *src,
                                 const float *src_coords, float *u, int t2, int time, int p_src_m,
                                 int p_src_M, int src_stride, int coord_stride, int order, void *stream);

/* fdtd_hip_source_inject_slab (below) on arrays padded by `order` cells per side: the scatter of one slab of
 * a domain cut along x, at any space order. */
int fdtd_hip_source_inject_slab_order(const struct fdtd_geom *g, const float *m, const float *src,
                                      const float *src_coords, float *u, int t2, int time, int p_src_m,
                                      int p_src_M, int src_stride, int coord_stride, int x_shift,
                                      int x_lo, int x_hi, int order, void *stream);

/* The two device kernels Kernel_HIP_Forward adds to the path, on device arrays: one damped sweep, and
 * the receiver interpolation of level t of u into rec[time][p_rec_m..p_rec_M].  The _order forms take
 * the space order (arrays padded by that many cells); the plain forms are order 4. */
int fdtd_hip_stencil_step_damped(const struct fdtd_geom *g, const float *m, const float *damp, float *u,
                                 int t0, int t1, int t2, void *stream);
int fdtd_hip_stencil_step_damped_order(const struct fdtd_geom *g, const float *m, const float *damp,
                                       float *u, int t0, int t1, int t2, int order, void *stream);
int fdtd_hip_receiver_sample(const struct fdtd_geom *g, const float *u, int t, const float *rec_coords,
                             float *rec, int time, int p_rec_m, int p_rec_M, int rec_stride,
                             int coord_stride, void *stream);
int fdtd_hip_receiver_sample_order(const struct fdtd_geom *g, const float *u, int t,
                                   const float *rec_coords, float *rec, int time, int p_rec_m,
                                   int p_rec_M, int rec_stride, int coord_stride, int order,
                                   void *stream);

/* One leapfrog update  u[t2] <- stencil(u[t0], u[t1], m)  over the ranges in g.
 * Replaces one launch of stencil_update_kernel_1step (cuda.cu:237-243) /
 * stencil_update_h100_scalar_pipelined_kernel (cuda_optimized.cu:404-416).
 * t0/t1/t2 are the ring indices (time%3, (time+2)%3, (time+1)%3; openacc.cpp:148).
 * u points at level 0 of the [3][nxp][nyp][nzp] array. */
int fdtd_hip_stencil_step(const struct fdtd_geom *g, const float *m, float *u,
                          int t0, int t1, int t2, int variant, void *stream);

/* Trilinear 8-corner scatter of sources p_src_m..p_src_M at time index `time` into
 * u[t2].  Replaces one launch of source_inject_kernel (cuda.cu:253-261).
 * src is [*][src_stride], src_coords is [*][coord_stride]. */
int fdtd_hip_source_inject(const struct fdtd_geom *g, const float *m,
                           const float *src, const float *src_coords, float *u,
                           int t2, int time, int p_src_m, int p_src_M,
                           int src_stride, int coord_stride, void *stream);

/* Slab form of the scatter, for a grid cut into slabs along x (the slowest axis) with one
 * slab per GPU.  The arrays hold the slab whose first interior plane is GLOBAL plane
 * x_shift; coordinates, origin and spacing stay global, so positions and weights are the
 * very numbers the single-domain call computes.  A corner is applied iff its global x index
 * lies in [x_lo, x_hi] (the y and z tests are those of openacc.cpp:192).  With x_shift = 0,
 * x_lo = x_m-1, x_hi = x_M+1 this is fdtd_hip_source_inject.  No counterpart in the reference
 * (single device only); the arithmetic is openacc.cpp:185-196. */
int fdtd_hip_source_inject_slab(const struct fdtd_geom *g, const float *m,
                                const float *src, const float *src_coords, float *u,
                                int t2, int time, int p_src_m, int p_src_M,
                                int src_stride, int coord_stride, int x_shift, int x_lo,
                                int x_hi, void *stream);

/* Two time steps (time and time + 1) in one call, on four separate level buffers of
 * nxp*nyp*nzp floats each: cur = u(time) and prev = u(time-1) are read, next1 =
 * u(time+1) and next2 = u(time+2) are written (interior cells only; the halo cells of
 * next1 and next2 keep their values and are read as such).  Both source scatters are
 * included.  When the shape allows (z extent a multiple of 4) the two
 * sweeps are fused into one kernel that moves 10 instead of 16 bytes per cell update;
 * otherwise they run one after the other.  Either way the result is bit for bit that of
 * two single steps.  This is the time fusion the reference driver has a knob for
 * (t_fuse, main.cpp:261,267) and no reference kernel implements.
 * next1 and next2 must not overlap each other, cur, prev or m (refused with
 * hipErrorInvalidValue): unlike the reference's ring, u(time+2) cannot be written over
 * u(time-1), because neighbouring workgroups still read it.
 * variant: FDTD_VARIANT_AUTO fuses boxes of at least 32 planes and 16 M cells whose
 * cross-section fills the sweep's tiles of 8 rows x 512 cells, or 16 x 256, to 50 %: smaller
 * boxes are faster with the single-step kernel, emptier tiles have not been measured.
 * FDTD_VARIANT_MARCH2 fuses every shape the kernel takes.  Any other variant selects the
 * single-step kernel used for both sweeps. */
int fdtd_hip_two_steps(const struct fdtd_geom *g, const float *m, const float *src,
                       const float *src_coords, const float *cur, const float *prev,
                       float *next1, float *next2, int time, int p_src_m, int p_src_M,
                       int src_stride, int coord_stride, int variant, void *stream);

/* fdtd_hip_two_steps_ex with the damping term of Kernel_HIP_Forward: damp is a device array of the padded
 * shape of m; u+ = dt^2 (lap - (-2 r1 u + r1 u-) m + r0 damp u) / (m + dt damp).  No reference arithmetic
 * (parity unpinned); where damp is 0 the values are those of fdtd_hip_two_steps.  Same shapes, same fall-back
 * to two single (damped) sweeps, same source handling; flags as for fdtd_hip_two_steps_ex. */
int fdtd_hip_two_steps_damped(const struct fdtd_geom *g, const float *m, const float *damp,
                              const float *src, const float *src_coords, const float *cur,
                              const float *prev, float *next1, float *next2, int time, int p_src_m,
                              int p_src_M, int src_stride, int coord_stride, int variant, int flags,
                              void *stream);

/* Would fdtd_hip_two_steps take the fused sweep for this box with nsrc sources (levels assumed
 * 16-byte aligned)?  1 or 0; variant as for fdtd_hip_two_steps. */
int fdtd_hip_two_steps_fuses(const struct fdtd_geom *g, int nsrc, int variant);

/* How the two-step sweep tiles a cross-section of ey rows of ez cells: *zstep = 512, 256 or 128 (uniform
 * z-tiles) or 0 (mixed: tiles of 512, then one of 256, then one of 128 cells), *ncuts = z-cuts per row,
 * *fill = cells / cells the tiles span.  Returns the workgroups per plane (0 for an empty cross-section). */
long fdtd_hip_two_steps_tiling(int ey, int ez, int *zstep, int *ncuts, double *fill);

/* Statistics: how many two-step sweep kernels this process has launched so far (tells a
 * caller whether its calls were fused or ran as two sweeps). */
long fdtd_hip_fused_launch_count(void);

/* Statistics: how many sweep-and-scatter launches (the one-launch-per-step time loop the
 * operators and fdtd_hip_time_loop use on small grids) this process has issued so far. */
long fdtd_hip_small_step_count(void);

/* The same with two extras for callers that measure or know more.  ev_begin / ev_end
 * (hipEvent_t, may be NULL): when the call is a single sweep launch they receive the start
 * and stop times of that dispatch itself, otherwise they are recorded on the stream before
 * the first and after the last launch of the call.  flags bit 0
 * (FDTD_TWO_STEPS_SOURCES_INSIDE) promises that no source corner lands in the halo layer,
 * which saves the two small launches that serve such corners. */
#define FDTD_TWO_STEPS_SOURCES_INSIDE 1
int fdtd_hip_two_steps_ex(const struct fdtd_geom *g, const float *m, const float *src,
                          const float *src_coords, const float *cur, const float *prev,
                          float *next1, float *next2, int time, int p_src_m, int p_src_M,
                          int src_stride, int coord_stride, int variant, int flags,
                          void *ev_begin, void *ev_end, void *stream);

/* Slab forms for domains cut along x (one slab per GPU; no counterpart in the reference, which
 * is single-device).  The level buffers and m are local arrays of (nx_loc+8) planes holding
 * planes x_shift-4 .. x_shift+nx_loc+3 of a domain of global_nx planes; g->x_m..x_M is the local
 * range to update (a subset of 0..nx_loc-1); source positions are global.  Halo-layer corners
 * are applied by the call whose range contains their plane (the first / last plane of the
 * domain also owns the plane beyond it).
 * two_steps: cur must be valid 4 planes and prev 2 planes beyond the range on either side
 * (ghost planes received from the neighbours, or the physical halo); u(time+1) is computed
 * two planes beyond the range as well, bit for bit as the owner of those planes computes it.
 * The fused sweep keeps those extra planes in registers.  Ranges it does not take (fewer than
 * 32 planes, other shapes) are run as two sweeps; with a `scratch` of at least
 * (x_M - x_m + 5) planes (nyp*nzp floats each) those build the extra planes there and store
 * nothing outside the range, which is what a range needs that is updated while its
 * neighbouring ranges are (the boundary zone of a slab); without scratch (NULL) the first of
 * the two sweeps stores u(time+1) on the extra planes of next1 too (same values as their owner
 * stores), so such calls must not run concurrently with calls on adjacent ranges.
 * flags: FDTD_TWO_STEPS_SOURCES_INSIDE as above; FDTD_TWO_STEPS_KEEP_LOW / _KEEP_HIGH make
 * every path store u(time+1) on the two planes before x_m / behind x_M (where the domain
 * has them).  A slab sets them on the side of a neighbour: its ghost planes of u(time+1),
 * which the next pair reads as u(t-1), are then computed here and need not be sent. */
#define FDTD_TWO_STEPS_KEEP_LOW 4
#define FDTD_TWO_STEPS_KEEP_HIGH 8
/* flags bits 8..15: n > 0 makes the last x-chunk of the fused sweep n planes shorter than the
 * others, so that its workgroups end early and free their CUs for a kernel that cannot run
 * beside the sweep (the message kernel of the ghost exchange); results do not depend on it. */
#define FDTD_TWO_STEPS_TAIL(n) (((n) & 0xff) << 8)
int fdtd_hip_two_steps_slab(const struct fdtd_geom *g, const float *m, const float *src,
                            const float *src_coords, const float *cur, const float *prev,
                            float *next1, float *next2, int time, int p_src_m, int p_src_M,
                            int src_stride, int coord_stride, int variant, int flags, int x_shift,
                            int global_nx, float *scratch, long scratch_floats, void *stream);
int fdtd_hip_one_step_slab(const struct fdtd_geom *g, const float *m, const float *src,
                           const float *src_coords, const float *cur, const float *prev, float *next,
                           int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                           int variant, int x_shift, int global_nx, void *stream);

/* The same decomposition as a session that owns the slabs on the devices, for callers that keep the
 * wavefield on the GPUs between calls (a 2048^3 grid is 139 GB of host arrays through the operator
 * above; here nothing of it exists on the host unless it is asked for).  No counterpart in the
 * reference (single device, main.cpp:53-72 takes host arrays); Kernel_HIP_MultiGPU is create +
 * set_sources + upload + advance (untimed steps) + advance (timed steps) + gather x 3 + destroy.
 *   create   geom describes the whole domain (padded extents, y / z ranges, spacings, origin; the x range
 *            must be the whole extent 0 .. nxp-9: the slabs cover it).  devices[0 .. ndev-1] are HIP device
 *            numbers, one slab each in this order along x; a device named twice carries two slabs (a
 *            rehearsal on fewer GPUs).  At least 8 planes per slab.  Allocates four level buffers and m per
 *            slab ((nx_slab + 8) planes each), three streams and the events of the schedule per slab, and
 *            starts one issuing thread per slab (FDTD_MULTIGPU_ONE_THREAD, or FDTD_MULTIGPU_THREADS=0 in the
 *            environment: the calling thread issues for every slab, device after device).
 *   set_sources  src[src_steps][src_stride], src_coords[coord_rows][coord_stride] (host), sources p_src_m ..
 *            p_src_M; copied to every device.  p_src_M < p_src_m: no sources.
 *   fill_standard_inputs  the driver's benchmark inputs (main.cpp:345-356: u = 0, m = 1.5, nsrc Ricker
 *            sources of nt samples at the positions of main.cpp:301-325) made on the devices; time index 0.
 *   upload   host u[3][nxp][nyp][nzp] (the reference's ring at time index time_m) and m[nxp][nyp][nzp]
 *            (NULL: keep the model already there) cut into the slabs.
 *   set_damping  the damping field of Kernel_HIP_Forward (host array of the padded shape of m; NULL: none): every
 *            later advance runs the damped update, two damped steps per sweep where the sweep is fused (parity
 *            unpinned, like Kernel_HIP_Forward).
 *   advance  steps time_m .. time_M inclusive; time_m must be the time index the session stands at (the
 *            one given to upload, 0 after fill, time_M + 1 of the previous advance).  Two steps per exchange,
 *            one single step at the end of an odd count.  Returns when every device has finished;
 *            timers (may be NULL) ->section0 += wall time of the call.
 *   gather   one ring slot (0 .. 2; slot t % 3 holds the newest level u(t), t = the session's time index)
 *            into a host array of nxp*nyp*nzp floats.
 *   max_abs  max |u| of the newest level and its number of NaN / Inf values, reduced on the devices.
 *   checksum the sum of all 32-bit words of one ring slot's level as a 64-bit integer (order-independent, so the
 *            same for every decomposition of the same run): compares a run on N slabs with the undivided one
 *            without a host copy of either.
 *   stats    see fdtd_multigpu.hip: slabs, operations, host issue and wait time per operation, schedule.
 * Every function returns 0 or a positive hipError_t; after a failed advance the session only accepts
 * destroy.  Calls on one session must not overlap. */
typedef struct fdtd_multigpu fdtd_multigpu;
#define FDTD_MULTIGPU_ONE_THREAD 1   /* issue from the calling thread only */
#define FDTD_MULTIGPU_NO_OVERLAP 2   /* one chain per slab: no interior / boundary split */
#define FDTD_MULTIGPU_NO_EXCHANGE 4  /* timing experiment: no pushes, no waits for neighbours; WRONG RESULTS */
int fdtd_hip_multigpu_create(const struct fdtd_geom *geom, int ndev, const int *devices, int flags,
                             fdtd_multigpu **out);
int fdtd_hip_multigpu_destroy(fdtd_multigpu *h);
int fdtd_hip_multigpu_set_sources(fdtd_multigpu *h, const float *src, int src_steps, int src_stride,
                                  const float *src_coords, int coord_rows, int coord_stride,
                                  int p_src_m, int p_src_M);
int fdtd_hip_multigpu_fill_standard_inputs(fdtd_multigpu *h, int nt, int nsrc);
int fdtd_hip_multigpu_upload(fdtd_multigpu *h, const float *u, const float *m, int time_m);
int fdtd_hip_multigpu_set_damping(fdtd_multigpu *h, const float *damp);
int fdtd_hip_multigpu_advance(fdtd_multigpu *h, int time_m, int time_M, struct profiler *timers);
int fdtd_hip_multigpu_gather(fdtd_multigpu *h, int ring_slot, float *host_level);
int fdtd_hip_multigpu_max_abs(fdtd_multigpu *h, float *max_abs, long *nonfinite);
int fdtd_hip_multigpu_checksum(fdtd_multigpu *h, int ring_slot, unsigned long long *sum);
int fdtd_hip_multigpu_stats(fdtd_multigpu *h, double out[10], int reset);

/* fdtd_hip_two_steps_slab without the work-space argument, for callers that cannot allocate
 * device memory: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference.  The call allocates and frees nothing (no
 * hipMalloc / hipFree, no allocating library routine); the work space of the two-sweep path is
 * an array of FDTD_STATIC_SCRATCH_FLOATS floats (16 MiB) inside the library's code object, one
 * copy per GPU, held until the process ends.  A range of more than that —
 * (x_M - x_m + 5) * nyp * nzp floats — is refused with hipErrorInvalidValue and a message on
 * stderr.  Results are those of fdtd_hip_two_steps_slab, bit for bit.  All calls on one GPU share
 * the array: calls that overlap in time on one GPU (different streams or threads) race.  The
 * first call on a device looks up the array's address there (hipGetSymbolAddress); make that
 * call outside graph capture. */
#define FDTD_STATIC_SCRATCH_FLOATS (4L << 20)
int fdtd_hip_two_steps_slab_noalloc(const struct fdtd_geom *g, const float *m, const float *src,
                                    const float *src_coords, const float *cur, const float *prev,
                                    float *next1, float *next2, int time, int p_src_m, int p_src_M,
                                    int src_stride, int coord_stride, int variant, int flags,
                                    int x_shift, int global_nx, void *stream);

/* One step (sweep of openacc.cpp:151-168 + scatter of openacc.cpp:173-203) on three separate
 * level buffers: next = u(time+1) from cur = u(time) and prev = u(time-1). */
int fdtd_hip_one_step(const struct fdtd_geom *g, const float *m, const float *src,
                      const float *src_coords, const float *cur, const float *prev, float *next,
                      int time, int p_src_m, int p_src_M, int src_stride, int coord_stride,
                      int variant, void *stream);

/* Copies the shell of one level buffer into another: the two layers of cells around the
 * updated box, which is every halo cell a sweep reads (openacc.cpp:161-163) or a scatter
 * touches (openacc.cpp:192); with_box != 0 copies the box as well.  In the reference's ring
 * of three, u(t+2) is written over u(t-1) and inherits its halo; a caller that rotates four
 * buffers through fdtd_hip_two_steps gets the same values by copying the shell of prev into
 * next2 before the call (unnecessary when all buffers hold the same halo, e.g. zeros, and
 * no source sits in the outermost cell layer). */
int fdtd_hip_copy_shell(const struct fdtd_geom *g, float *dst_level, const float *src_level,
                        int with_box, void *stream);

/* The whole time loop on device-resident fields: steps time_m..time_M, ring
 * indexing and section timers exactly as the operators do (timers may be NULL, in
 * which case nothing is timed).  This is the body of every Kernel_* operator between
 * its host-to-device and device-to-host copies (cuda.cu:232-314).  The call returns
 * after a final stream synchronise.  Beyond that: where it takes the two-step sweep
 * (boxes of 16 M cells and more) it uses a fourth level buffer, allocated at the first such
 * call on a device and kept until fdtd_hip_release_device_copies, reads the source positions back once (one small device-to-host
 * copy and a stream synchronise before the first step) and ends with up to four
 * level copies that put the three newest levels into their ring slots; where it takes
 * the one-launch-per-step loop (smaller boxes) it allocates a table of the sources and
 * reads the positions back the same way.  Callers that cannot have allocations or
 * synchronisations inside their loop drive fdtd_hip_two_steps / fdtd_hip_one_step
 * (or the sweep and scatter calls) themselves. */
int fdtd_hip_time_loop(const struct fdtd_geom *g, const float *m, const float *src,
                       const float *src_coords, float *u, int p_src_m, int p_src_M,
                       int src_stride, int coord_stride, int time_m, int time_M,
                       int variant, void *stream, struct profiler *timers);

int fdtd_hip_time_loop_order(const struct fdtd_geom *g, const float *m, const float *src,
                             const float *src_coords, float *u, int p_src_m, int p_src_M,
                             int src_stride, int coord_stride, int time_m, int time_M,
                             int variant, int space_order, void *stream, struct profiler *timers);

/* fp16-storage counterparts (u is __half[3][nxp][nyp][nzp]; m, src stay fp32). */
int fdtd_hip_stencil_step_h16(const struct fdtd_geom *g, const float *m, void *u_h16,
                              int t0, int t1, int t2, void *stream);
int fdtd_hip_source_inject_h16(const struct fdtd_geom *g, const float *m,
                               const float *src, const float *src_coords,
                               void *u_h16, int t2, int time, int p_src_m,
                               int p_src_M, int src_stride, int coord_stride,
                               void *stream);
int fdtd_hip_convert_f32_to_h16(const float *in, void *out_h16, size_t n, void *stream);
int fdtd_hip_convert_h16_to_f32(const void *in_h16, float *out, size_t n, void *stream);

/* ---------------------------------------------------------------------------
 * Host-side input generators and parity metrics of the reference driver, restated
 * so that a caller can reproduce its runs without the driver.
 * ------------------------------------------------------------------------- */

/* main.cpp:290-298: src[t*S+s] = (1-2a)exp(-a), a = pi^2 f0^2 (t*dt - 1/f0)^2,
 * f0 = 10, all in float. */
void fdtd_fill_ricker(float *src, int T, int S, float dt);
/* main.cpp:301-325: the first S of the 27 tick combinations {1/4,1/2,3/4}*(n-1)*0.1
 * per axis (x-major), then the grid centre for any further sources. */
void fdtd_fill_source_coords(float *coords, int S, int nx, int ny, int nz);
/* main.cpp:525-532: level0 = level1 = sin(i*0.001f)*10+100 over all padded cells,
 * level2 = 0, m = 1.5.  u has 3*volp floats, m has volp floats. */
void fdtd_fill_correctness_inputs(float *u, float *m, size_t volp);

/* main.cpp:573-593: NaN/Inf in `test` are counted and skipped; the rest feed
 * max_abs, max_rel (where |ref| > 1e-10) and L2 = sqrt(sum d^2/(sum ref^2+1e-30)). */
struct fdtd_parity {
  double max_abs, max_rel, l2;
  long nan_count, inf_count;
};
void fdtd_compare_fields(const float *test, const float *ref, size_t n,
                         struct fdtd_parity *out);

/* Convenience runners with the standard benchmark inputs of main.cpp:279-399
 * (m = 1.5, u = 0, h = 0.1, dt = 1e-3, origin 0, nsrc Ricker sources, time 0..nt-1).
 * They allocate and fill host buffers, call the matching Kernel_* operator, report
 * max|u| over the three levels and the timers, and free everything.
 * u_out may be NULL; if not it receives 3*(NX+8)*(NY+8)*(NZ+8) floats. */
int run_openacc(int NX, int NY, int NZ, int nt, int nsrc, float *u_out,
                float *max_abs_u, struct profiler *timers);
int run_cuda(int NX, int NY, int NZ, int nt, int nsrc, float *u_out,
             float *max_abs_u, struct profiler *timers);
int run_cuda_optimized(int NX, int NY, int NZ, int nt, int nsrc, float *u_out,
                       float *max_abs_u, struct profiler *timers);

/* Version string of the library ("fdtd-hip gfx950 <variant list>"). */
const char *fdtd_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FDTD_ABI_H */
