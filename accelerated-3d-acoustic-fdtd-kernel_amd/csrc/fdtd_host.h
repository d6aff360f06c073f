// fdtd_host.h — host-side helpers shared by the operator wrappers (fdtd_operators.hip) and the
// multi-device operator (fdtd_multigpu.hip).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "fdtd_device.h"

// Shape checks of the operators' arguments (fdtd_operators.hip): 0 or hipErrorInvalidValue.
int fdtd_check_operator_args(const dataobj *m_vec, const dataobj *src_vec, const dataobj *coords_vec,
                             const dataobj *u_vec, int p_src_M, int p_src_m);

// Do the three ring levels of a host array u[3][nxp][nyp][nzp] carry bit-identical shells (the two layers of
// cells around the updated box)?  Then no shell has to follow the ring when four level buffers rotate.
inline bool fdtd_host_shells_identical(const float *u, size_t vol, const fdtd_geom &g) {
  const long sx = (long)g.nyp * g.nzp, sy = g.nzp;
  const int ez = g.z_M - g.z_m + 1;
  for (int x = g.x_m - 2; x <= g.x_M + 2; ++x)
    for (int y = g.y_m - 2; y <= g.y_M + 2; ++y) {
      const bool whole = x < g.x_m || x > g.x_M || y < g.y_m || y > g.y_M;
      const long row = (long)(x + FDTD_HALO) * sx + (long)(y + FDTD_HALO) * sy + FDTD_HALO + g.z_m - 2;
      for (int l = 1; l < 3; ++l) {
        const float *a = u + row, *b = u + (size_t)l * vol + row;
        if (whole) {
          if (memcmp(a, b, (size_t)(ez + 4) * sizeof(float))) return false;
        } else if (memcmp(a, b, 2 * sizeof(float)) || memcmp(a + ez + 2, b + ez + 2, 2 * sizeof(float))) {
          return false;
        }
      }
    }
  return true;
}

// Can a corner of the sources p_src_m .. p_src_M land in the halo layer of the box?  (Positions as
// openacc.cpp:185-187 computes them; coords is a host array.)
inline bool fdtd_host_sources_inside(const float *coords, int coord_stride, int p_src_m, int p_src_M, const fdtd_geom &g) {
  for (int p = p_src_m; p <= p_src_M; ++p) {
    const int posx = (int)floorf((-g.o_x + coords[(size_t)p * coord_stride + 0]) / g.h_x);
    const int posy = (int)floorf((-g.o_y + coords[(size_t)p * coord_stride + 1]) / g.h_y);
    const int posz = (int)floorf((-g.o_z + coords[(size_t)p * coord_stride + 2]) / g.h_z);
    if (posx < g.x_m || posx + 1 > g.x_M || posy < g.y_m || posy + 1 > g.y_M || posz < g.z_m || posz + 1 > g.z_M)
      return false;
  }
  return true;
}

// Four rotating level buffers back into ring order: the levels at pos[0 .. 2] go to the buffers tgt[0 .. 2]
// (indices into P), each with its box and shell (fdtd_hip_copy_shell), through the free buffer.
inline int fdtd_ring_settle(const fdtd_geom *g, float *const P[4], const int pos_in[3], int free_pos, const int tgt[3],
                            void *stream) {
  int pos[3] = {pos_in[0], pos_in[1], pos_in[2]};
  for (int it = 0; it < 8; ++it) {
    int mover = -1;
    for (int r = 0; r < 3; ++r)
      if (pos[r] != tgt[r] && tgt[r] == free_pos) mover = r;
    if (mover < 0)
      for (int r = 0; r < 3; ++r)
        if (pos[r] != tgt[r]) mover = r;
    if (mover < 0) break;
    const int rc = fdtd_hip_copy_shell(g, P[free_pos], P[pos[mover]], 1, stream);
    if (rc) return rc;
    const int old = pos[mover];
    pos[mover] = free_pos;
    free_pos = old;
  }
  return 0;
}

// Transfers between the caller's pageable arrays and the device.  A plain hipMemcpy of pageable
// memory runs at 27 GB/s towards the device and 17 GB/s back on an MI355X host, and the caller's
// arrays cannot be page-locked in place (hipHostRegister refuses them there), so large transfers
// go through three pinned 16 MB bounce buffers: a few host threads copy one chunk while the copy
// engine moves the previous ones (54 GB/s either way, the rate of one pinned transfer).  The
// reference's wrappers use plain cudaMemcpy (cuda.cu:204-214,317); the result is the same bytes.
// FDTD_STAGING=0 selects plain hipMemcpy, FDTD_STAGE_THREADS the number of host threads (4);
// FDTD_STAGE_CHUNK_KB and FDTD_STAGE_MIN_KB change the chunk size and the smallest staged
// transfer (tests use them to run many small chunks).
class StagedCopier {
 public:
  static constexpr int NBUF = 3;

  StagedCopier() {
    const char *e = getenv("FDTD_STAGING");
    if (e && atoi(e) == 0) return;
    const char *t = getenv("FDTD_STAGE_THREADS");
    threads_ = t ? atoi(t) : 4;
    if (threads_ < 1) threads_ = 1;
    if (threads_ > 16) threads_ = 16;
    const char *c = getenv("FDTD_STAGE_CHUNK_KB"), *lo = getenv("FDTD_STAGE_MIN_KB");
    if (c && atol(c) > 0) CHUNK = (size_t)atol(c) << 10;
    if (lo && atol(lo) >= 0) MIN_BYTES = (size_t)atol(lo) << 10;
    enabled_ = true;
  }
  ~StagedCopier() { release(); }

  int to_device(void *dst, const void *src, size_t bytes) { return run(dst, src, bytes, true); }
  int to_host(void *dst, const void *src, size_t bytes) { return run(dst, src, bytes, false); }

 private:
  size_t CHUNK = size_t(16) << 20;
  size_t MIN_BYTES = size_t(32) << 20;   // below this a plain copy is as fast
  bool enabled_ = false, ready_ = false;
  int threads_ = 4;
  char *buf_[NBUF] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_[NBUF] = {nullptr, nullptr, nullptr};
  hipStream_t stream_ = nullptr;

  void release() {
    for (int i = 0; i < NBUF; ++i) {
      if (buf_[i]) (void)hipHostFree(buf_[i]);
      if (ev_[i]) (void)hipEventDestroy(ev_[i]);
      buf_[i] = nullptr;
      ev_[i] = nullptr;
    }
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
    ready_ = false;
  }
  bool prepare() {
    if (ready_) return true;
    bool ok = hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < NBUF; ++i)
      ok = hipHostMalloc((void **)&buf_[i], CHUNK, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      release();
      enabled_ = false;   // no pinned memory to be had: plain copies from here on
    }
    ready_ = ok;
    return ok;
  }
  void host_copy(char *dst, const char *src, size_t len) const {
    if (threads_ == 1 || len < (size_t(1) << 20)) {
      memcpy(dst, src, len);
      return;
    }
    const size_t per = ((len + threads_ - 1) / threads_ + 4095) & ~size_t(4095);
    std::vector<std::thread> pool;
    for (size_t lo = per; lo < len; lo += per)
      pool.emplace_back([=] { memcpy(dst + lo, src + lo, len - lo < per ? len - lo : per); });
    memcpy(dst, src, per < len ? per : len);
    for (std::thread &t : pool) t.join();
  }
  int run(void *dst, const void *src, size_t bytes, bool h2d) {
    if (!enabled_ || bytes < MIN_BYTES || !prepare()) {
      FDTD_HIP_CHECK(hipMemcpy(dst, src, bytes, h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
      return 0;
    }
    FDTD_HIP_CHECK(hipDeviceSynchronize());   // whatever produced the source has finished
    const size_t nch = (bytes + CHUNK - 1) / CHUNK;
    auto len_of = [&](size_t c) { return bytes - c * CHUNK < CHUNK ? bytes - c * CHUNK : CHUNK; };
    if (h2d) {
      for (size_t c = 0; c < nch; ++c) {
        const int b = (int)(c % NBUF);
        if (c >= (size_t)NBUF) FDTD_HIP_CHECK(hipEventSynchronize(ev_[b]));   // the buffer's last transfer
        host_copy(buf_[b], (const char *)src + c * CHUNK, len_of(c));
        FDTD_HIP_CHECK(hipMemcpyAsync((char *)dst + c * CHUNK, buf_[b], len_of(c), hipMemcpyHostToDevice, stream_));
        FDTD_HIP_CHECK(hipEventRecord(ev_[b], stream_));
      }
    } else {
      // chunk c is on its way into a buffer while chunk c - (NBUF - 1) is copied out of another
      for (size_t c = 0; c < nch + NBUF - 1; ++c) {
        if (c < nch) {
          const int b = (int)(c % NBUF);
          FDTD_HIP_CHECK(hipMemcpyAsync(buf_[b], (const char *)src + c * CHUNK, len_of(c), hipMemcpyDeviceToHost,
                                        stream_));
          FDTD_HIP_CHECK(hipEventRecord(ev_[b], stream_));
        }
        if (c >= (size_t)(NBUF - 1)) {
          const size_t cc = c - (NBUF - 1);
          const int b = (int)(cc % NBUF);
          FDTD_HIP_CHECK(hipEventSynchronize(ev_[b]));
          host_copy((char *)dst + cc * CHUNK, buf_[b], len_of(cc));
        }
      }
    }
    FDTD_HIP_CHECK(hipStreamSynchronize(stream_));
    return 0;
  }
};

// The copier of the current device: its pinned bounce buffers, stream and events are made once per device
// and kept for the life of the process (48 MB of page-locked memory per device in use; never freed, so that
// nothing of it is torn down after the HIP runtime at exit).  Operator calls are sequential (main.cpp:394);
// callers serialise their use of it.
inline StagedCopier &staged_copier_for_current_device() {
  static StagedCopier *per_device[64] = {};
  static std::string settings[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  // the switches a copier reads when it is made; a copier made under other settings is replaced
  std::string now;
  for (const char *name : {"FDTD_STAGING", "FDTD_STAGE_THREADS", "FDTD_STAGE_CHUNK_KB", "FDTD_STAGE_MIN_KB"}) {
    const char *v = getenv(name);
    now += v ? v : "";
    now += '|';
  }
  if (per_device[dev] && settings[dev] != now) {
    delete per_device[dev];
    per_device[dev] = nullptr;
  }
  if (!per_device[dev]) {
    per_device[dev] = new StagedCopier();
    settings[dev] = now;
  }
  return *per_device[dev];
}
