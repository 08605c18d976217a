// Internal: device-wide int64 exclusive scan used by the tile censuses and row-offset builders.
#pragma once
#include "common.h"

// bytes of scratch (int64 partials of every level) a scan over n items needs
size_t bnpk_scan_scratch_bytes(int64_t n);

// d_out[i] = sum_{j<i} f(d_in[j]) with f(v) = window > 1 ? max(0, v-(window-1)) : v.
// In-place (d_out == d_in) is allowed.  If write_total, d_out has n+1 entries and d_out[n] = total.
int bnpk_scan_launch(bnpk_ctx* ctx, const int64_t* d_in, int64_t n, int window, int64_t* d_out,
                     bool write_total, int64_t* d_scratch, hipStream_t stream);
