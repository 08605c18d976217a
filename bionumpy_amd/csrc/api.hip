// Context, diagnostics, host staging and per-kernel timers of the bnpk C-ABI (include/bnpk.h).
#include <errno.h>
#include <string.h>
#include <unistd.h>

#include <thread>
#include <vector>

#include "common.h"

extern "C" {

int bnpk_version(void) { return 100; }

const char* bnpk_strerror(int status) {
  switch (status) {
    case BNPK_OK: return "ok";
    case BNPK_ERR_ARG: return "bad argument";
    case BNPK_ERR_ALIGN: return "device byte buffer is not 16-byte aligned";
    case BNPK_ERR_HIP: return "HIP runtime error";
    case BNPK_ERR_NOMEM: return "out of device memory / workspace too small";
    case BNPK_ERR_NODEVICE: return "no HIP device visible";
    case BNPK_ERR_RANGE: return "size out of supported range";
    default: return "unknown bnpk status";
  }
}

int bnpk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bnpk_ctx_create(int device, bnpk_ctx** out) {
  if (!out) return BNPK_ERR_ARG;
  int n = bnpk_device_count();
  if (n <= 0 || device < 0 || device >= n) return BNPK_ERR_NODEVICE;
  bnpk_ctx* ctx = new bnpk_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return BNPK_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->compute_units = prop.multiProcessorCount;
  *out = ctx;
  return BNPK_OK;
}

void bnpk_ctx_destroy(bnpk_ctx* ctx) {
  if (!ctx) return;
  for (auto& p : ctx->pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->scratch_event) (void)hipEventDestroy(ctx->scratch_event);
  delete ctx;
}

const char* bnpk_last_hip_error(bnpk_ctx* ctx) {
  return hipGetErrorString(ctx ? ctx->last_err : hipSuccess);
}

int bnpk_device_info(bnpk_ctx* ctx, char* name64, int* compute_units, int64_t* hbm_bytes) {
  if (!ctx) return BNPK_ERR_ARG;
  hipDeviceProp_t prop;
  BNPK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name64) {
    // gcnArchName carries the ISA ("gfx950:sramecc+:xnack-"), name the marketing string
    snprintf(name64, 64, "%s", prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return BNPK_OK;
}

// ---- timers ---------------------------------------------------------------------------------
int bnpk_prof_enable(bnpk_ctx* ctx, int on) {
  if (!ctx) return BNPK_ERR_ARG;
  ctx->prof = on != 0;
  return BNPK_OK;
}

static int resolve_pending(bnpk_ctx* ctx) {
  for (auto& p : ctx->pending) {
    BNPK_HIP(ctx, hipEventSynchronize(p.stop));
    float ms = 0.f;
    BNPK_HIP(ctx, hipEventElapsedTime(&ms, p.start, p.stop));
    ctx->entries[p.entry].total_ms += ms;
    ctx->entries[p.entry].launches += 1;
    ctx->event_pool.push_back(p.start);
    ctx->event_pool.push_back(p.stop);
  }
  ctx->pending.clear();
  return BNPK_OK;
}

int bnpk_prof_reset(bnpk_ctx* ctx) {
  if (!ctx) return BNPK_ERR_ARG;
  BNPK_CHECK(resolve_pending(ctx));
  ctx->entries.clear();
  return BNPK_OK;
}

int bnpk_prof_count(bnpk_ctx* ctx) {
  if (!ctx) return BNPK_ERR_ARG;
  int s = resolve_pending(ctx);
  if (s != BNPK_OK) return s;
  return (int)ctx->entries.size();
}

int bnpk_prof_get(bnpk_ctx* ctx, int i, char* name64, double* total_ms, int64_t* launches) {
  if (!ctx || i < 0 || i >= (int)ctx->entries.size()) return BNPK_ERR_ARG;
  const auto& e = ctx->entries[i];
  if (name64) snprintf(name64, 64, "%s", e.name.c_str());
  if (total_ms) *total_ms = e.total_ms;
  if (launches) *launches = e.launches;
  return BNPK_OK;
}

int bnpk_set_option(bnpk_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return BNPK_ERR_ARG;
  if (!strcmp(name, "finish_mode")) {
    if (value < 0 || value > 6) return BNPK_ERR_ARG;
    ctx->finish_mode = (int)value;
    return BNPK_OK;
  }
  if (!strcmp(name, "fastq_encoder")) {
    if (value < 0 || value > 1) return BNPK_ERR_ARG;
    ctx->fastq_encoder = (int)value;
    return BNPK_OK;
  }
  if (!strcmp(name, "index_pairs")) {
    if (value < 0 || value > 1) return BNPK_ERR_ARG;
    ctx->index_pairs = (int)value;
    return BNPK_OK;
  }
  if (!strcmp(name, "sparse_claim")) {
    if (value < 0 || value > 1) return BNPK_ERR_ARG;
    ctx->sparse_claim = (int)value;
    return BNPK_OK;
  }
  if (!strcmp(name, "l1_ring")) {
    if (value < 0 || value > 1) return BNPK_ERR_ARG;
    ctx->l1_ring = (int)value;
    return BNPK_OK;
  }
  return BNPK_ERR_ARG;
}

// ---- host staging -----------------------------------------------------------------------------
int bnpk_host_alloc(size_t bytes, void** h_out) {
  if (!h_out) return BNPK_ERR_ARG;
  if (hipHostMalloc(h_out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();                             // the failure is reported as a status; it must not stick to the runtime
    *h_out = nullptr;
    return BNPK_ERR_NOMEM;
  }
  return BNPK_OK;
}

int bnpk_host_free(void* h_ptr) {
  if (h_ptr && hipHostFree(h_ptr) != hipSuccess) return BNPK_ERR_HIP;
  return BNPK_OK;
}

int bnpk_copy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream) {
  if (bytes == 0) return BNPK_OK;
  if (!d_dst || !h_src) return BNPK_ERR_ARG;
  if (hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return BNPK_ERR_HIP;
  return BNPK_OK;
}

// A plain file (page cache or disk) into a page-locked buffer by `n_threads` threads, each with a slice of its own, and on
// to the device piece by piece while the rest is still being read — the Python reader did this with a thread pool and spent
// its time on the interpreter lock once the threads also had copies to start.
int bnpk_pread_parallel(bnpk_ctx* ctx, int fd, int64_t file_offset, void* h_dst, int64_t bytes, int n_threads, int64_t piece_bytes,
                        void* d_dst, void* stream, int64_t* h_read) {
  if (fd < 0 || file_offset < 0 || bytes < 0 || n_threads < 1 || piece_bytes < 1 || !h_read || (bytes > 0 && !h_dst)) return BNPK_ERR_ARG;
  if (d_dst && !ctx) return BNPK_ERR_ARG;
  *h_read = 0;
  if (bytes == 0) return BNPK_OK;
  n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, bytes / piece_bytes));
  const int64_t step = ceil_div(ceil_div(bytes, (int64_t)n_threads), (int64_t)4096) * 4096;      // page-aligned slices
  std::vector<int64_t> got(n_threads, 0);
  std::vector<int> status(n_threads, BNPK_OK);
  auto work = [&](int t) {
    if (d_dst && hipSetDevice(ctx->device) != hipSuccess) { status[t] = BNPK_ERR_HIP; return; }
    int64_t a = (int64_t)t * step;
    const int64_t b = std::min(a + step, bytes);
    while (a < b) {
      const int64_t e = std::min(b, a + piece_bytes);
      int64_t at = a;
      while (at < e) {
        const ssize_t n = pread(fd, (char*)h_dst + at, (size_t)(e - at), (off_t)(file_offset + at));
        if (n < 0 && errno == EINTR) continue;
        if (n <= 0) break;                                   // error, or the file ends here
        at += n;
      }
      if (at > a && d_dst &&
          hipMemcpyAsync((char*)d_dst + a, (const char*)h_dst + a, (size_t)(at - a), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        status[t] = BNPK_ERR_HIP;
        return;
      }
      got[t] += at - a;
      if (at < e) return;                                    // short: nothing behind it is read
      a = e;
    }
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
  work(0);
  for (auto& th : threads) th.join();
  // the bytes read are the contiguous prefix: a short slice ends the count
  int64_t total = 0;
  for (int t = 0; t < n_threads; ++t) {
    if (status[t] != BNPK_OK) return status[t];
    total += got[t];
    if (got[t] < std::min(step, bytes - (int64_t)t * step)) break;
  }
  *h_read = total;
  return BNPK_OK;
}

// How many times `byte` occurs in [file_offset, file_offset + bytes) of the open file: pread into a buffer per thread and a
// counting loop the compiler vectorises.  What a reader that begins in the middle of a file needs to say which LINE of the
// file it is at (FormatException.line_number counts from the start of the file: bionumpy/io/parser.py:141-143).
int bnpk_count_byte_file(int fd, int64_t file_offset, int64_t bytes, uint8_t byte, int n_threads, int64_t* h_count) {
  if (fd < 0 || file_offset < 0 || bytes < 0 || n_threads < 1 || !h_count) return BNPK_ERR_ARG;
  *h_count = 0;
  if (bytes == 0) return BNPK_OK;
  const int64_t piece = 4 << 20;
  n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, bytes / piece));
  const int64_t step = ceil_div(ceil_div(bytes, (int64_t)n_threads), (int64_t)4096) * 4096;
  std::vector<int64_t> found(n_threads, 0);
  std::vector<int> status(n_threads, BNPK_OK);
  auto work = [&](int t) {
    std::vector<uint8_t> buf((size_t)std::min(piece, step));
    int64_t a = (int64_t)t * step;
    const int64_t b = std::min(a + step, bytes);
    int64_t c = 0;
    while (a < b) {
      const int64_t want = std::min<int64_t>(b - a, (int64_t)buf.size());
      const ssize_t n = pread(fd, buf.data(), (size_t)want, (off_t)(file_offset + a));
      if (n < 0 && errno == EINTR) continue;
      if (n <= 0) { status[t] = BNPK_ERR_ARG; return; }      // the range reaches behind the end of the file
      const uint8_t* p = buf.data();
      int64_t k = 0;
      for (ssize_t i = 0; i < n; ++i) k += p[i] == byte;
      c += k;
      a += n;
    }
    found[t] = c;
  };
  std::vector<std::thread> threads;
  for (int t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
  work(0);
  for (auto& th : threads) th.join();
  int64_t total = 0;
  for (int t = 0; t < n_threads; ++t) {
    if (status[t] != BNPK_OK) return status[t];
    total += found[t];
  }
  *h_count = total;
  return BNPK_OK;
}

int bnpk_copy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream) {
  if (bytes == 0) return BNPK_OK;
  if (!h_dst || !d_src) return BNPK_ERR_ARG;
  if (hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    return BNPK_ERR_HIP;
  return BNPK_OK;
}

int bnpk_fetch_i64(bnpk_ctx* ctx, const int64_t* d_src, int64_t n, int64_t* h_dst, void* stream) {
  if (!ctx || n < 0 || n > 4096) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_src || !h_dst) return BNPK_ERR_ARG;
  if (!ctx->mailbox) BNPK_HIP(ctx, hipHostMalloc(&ctx->mailbox, 4096 * sizeof(int64_t), hipHostMallocDefault));
  hipStream_t s = (hipStream_t)stream;
  BNPK_HIP(ctx, hipMemcpyAsync(ctx->mailbox, d_src, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  memcpy(h_dst, ctx->mailbox, (size_t)n * 8);
  return BNPK_OK;
}

int bnpk_stream_sync(void* stream) {
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BNPK_ERR_HIP;
  return BNPK_OK;
}

}  // extern "C"

// ---- internals ----------------------------------------------------------------------------------
int bnpk_scratch(bnpk_ctx* ctx, size_t bytes, void** out, hipStream_t stream) {
  // One arena, any number of streams: a call that arrives on another stream than the arena's last user first waits
  // (on the device) for everything that user has enqueued.  A ctx serves one host thread at a time.
  if (ctx->scratch_in_use && ctx->scratch_stream != stream) {
    if (!ctx->scratch_event) BNPK_HIP(ctx, hipEventCreateWithFlags(&ctx->scratch_event, hipEventDisableTiming));
    BNPK_HIP(ctx, hipEventRecord(ctx->scratch_event, ctx->scratch_stream));
    BNPK_HIP(ctx, hipStreamWaitEvent(stream, ctx->scratch_event, 0));
  }
  ctx->scratch_stream = stream;
  ctx->scratch_in_use = true;
  if (bytes > ctx->scratch_bytes) {
    // grow-only; hipFree synchronises the device, so earlier users of the old arena are done
    if (ctx->scratch) BNPK_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = bytes + (bytes >> 2) + (1 << 20);
    if (hipMalloc(&ctx->scratch, want) != hipSuccess) {
      (void)hipGetLastError();
      ctx->scratch = nullptr;
      return BNPK_ERR_NOMEM;
    }
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return BNPK_OK;
}

bnpk_timer::bnpk_timer(bnpk_ctx* c, const char* name, hipStream_t s) : ctx(c), stream(s) {
  if (!ctx || !ctx->prof) return;
  for (size_t i = 0; i < ctx->entries.size(); ++i)
    if (ctx->entries[i].name == name) { entry = (int)i; break; }
  if (entry < 0) {
    ctx->entries.push_back(bnpk_prof_entry{name});
    entry = (int)ctx->entries.size() - 1;
  }
  auto take = [&](hipEvent_t* e) {
    if (!ctx->event_pool.empty()) { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return true; }
    return hipEventCreate(e) == hipSuccess;
  };
  if (!take(&start) || !take(&stop)) { entry = -1; return; }
  (void)hipEventRecord(start, stream);
}

bnpk_timer::~bnpk_timer() {
  if (entry < 0) return;
  (void)hipEventRecord(stop, stream);
  ctx->pending.push_back(bnpk_pending_event{entry, start, stop});
}
