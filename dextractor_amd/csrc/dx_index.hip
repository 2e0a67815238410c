// dx_index.hip -- GPU text front end for .quiva images.
//
// Replaces the host's fgets/strlen walk (Read_Lines QV.c:751-798 and the structural checks of
// QVcoding_Scan QV.c:948-978) for large files: a newline scan gives every line's start, one
// thread per entry then checks the 6-line structure (header starts with '@' and holds a '/', five
// data lines of equal length) and emits the entry index (offset of the first data line, symbols
// per line).  Only the header lines travel back to the host, where sscanf parses the four fields
// exactly as the reference does.  Errors are reported as the reference would meet them: the
// lowest offending line number wins.
//
// Roofline: HBM, 1 byte read per text byte twice (count pass + fill pass).
#include "dx_internal.hpp"
#include "dx_device.hpp"

#define IDX_PER_THREAD 64                          // text bytes per thread (4 x 16-byte chunks)
#define IDX_TILE       (DX_BLOCK * IDX_PER_THREAD) // 16 KiB per workgroup

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *p, uint64_t pos, uint64_t n)
{ if (pos >= n) return 0;
  const int valid = n - pos >= 16 ? 16 : (int) (n - pos);
  return chunk_eq_mask(load_chunk(p + pos, valid), '\n') & ((1u << valid) - 1u);
}

__global__ __launch_bounds__(DX_BLOCK)
void k_nl_count(const uint8_t *text, uint64_t n, uint32_t *tile_cnt)
{ __shared__ uint32_t s_w[DX_WAVES_PER_BLK];
  const uint64_t base = (uint64_t) blockIdx.x * IDX_TILE + (uint64_t) threadIdx.x * IDX_PER_THREAD;
  uint32_t c = 0;
  #pragma unroll
  for (int k = 0; k < IDX_PER_THREAD / 16; k++)
    c += __popc(nl_mask16(text, base + 16 * k, n));
  const uint32_t w = wave_sum(c);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0)
    { uint32_t t = 0;
      for (int k = 0; k < DX_WAVES_PER_BLK; k++) t += s_w[k];
      tile_cnt[blockIdx.x] = t;
    }
}

// line_start[0] = 0; line_start[k+1] = position after the k-th newline
__global__ __launch_bounds__(DX_BLOCK)
void k_nl_fill(const uint8_t *text, uint64_t n, const uint64_t *tile_off, uint64_t *line_start)
{ __shared__ uint32_t s_w[DX_WAVES_PER_BLK];
  const uint64_t base = (uint64_t) blockIdx.x * IDX_TILE + (uint64_t) threadIdx.x * IDX_PER_THREAD;
  uint32_t m[IDX_PER_THREAD / 16], c = 0;
  #pragma unroll
  for (int k = 0; k < IDX_PER_THREAD / 16; k++)
    { m[k] = nl_mask16(text, base + 16 * k, n);
      c   += __popc(m[k]);
    }
  const uint32_t incl = wave_incl_scan(c);
  if (lane_id() == 63) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t pre = 0;
  for (int k = 0; k < (int) (threadIdx.x >> 6); k++) pre += s_w[k];
  uint64_t idx = tile_off[blockIdx.x] + pre + incl - c + 1;       // +1: slot 0 is the file start
  #pragma unroll
  for (int k = 0; k < IDX_PER_THREAD / 16; k++)
    { uint32_t mm = m[k];
      while (mm)
        { const int b = __ffs(mm) - 1;
          line_start[idx++] = base + 16 * k + b + 1;
          mm &= mm - 1u;
        }
    }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    line_start[0] = 0;
}

// one thread per entry: structure checks in the reference's order, entry index, header extent
__global__ __launch_bounds__(DX_BLOCK)
void k_qv_entries(const uint8_t *text, const uint64_t *line_start, uint64_t nlines, uint64_t nent,
                  uint64_t *off, uint32_t *len, uint32_t *hdr_len, unsigned long long *err)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i >= nent) return;
  const uint64_t l0 = 6 * i;                                       // 0-based index of the header line
  const uint64_t hs = line_start[l0], he = line_start[l0 + 1] - 1; // header without its newline
  const uint64_t hl = he - hs;
  hdr_len[i] = (uint32_t) (hl > 0xffffffu ? 0xffffffu : hl);
  off[i] = line_start[l0 + 1];
  len[i] = 0;
  #define FAIL(line1, code) { atomicMin(err, ((unsigned long long) (line1) << 8) | (code)); return; }
  if (hl == 0 || text[hs] != '@') FAIL(l0 + 1, DX_IDX_NO_HEADER)   // QV.c:954
  { bool slash = false;                                            // QV.c:958: index(Read+1,'/')
    for (uint64_t k = hs + 1; k < he && !slash; k++)
      slash = text[k] == '/';
    if (!slash) FAIL(l0 + 1, DX_IDX_BAD_HEADER)
  }
  const uint64_t have = nlines - l0 - 1 < 5 ? nlines - l0 - 1 : 5; // data lines present
  uint64_t first = 0;
  for (uint64_t j = 0; j < have; j++)                              // QV.c:785-796
    { const uint64_t ll = line_start[l0 + 2 + j] - 1 - line_start[l0 + 1 + j];
      if (j == 0) first = ll;
      else if (ll != first) FAIL(l0 + 2 + j, DX_IDX_RAGGED)
    }
  if (have < 5) FAIL(nlines + 1, DX_IDX_INCOMPLETE)                // QV.c:788-790
  if (first > 0x7fffffffull) FAIL(l0 + 6, DX_IDX_TOO_LONG)
  len[i] = (uint32_t) first;
  #undef FAIL
}

// header lines packed back to back (each followed by '\n') for the host's sscanf
__global__ __launch_bounds__(DX_BLOCK)
void k_gather_headers(const uint8_t *text, const uint64_t *line_start, uint64_t nent,
                      const uint64_t *hdr_pos, uint8_t *blob)
{ const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  for (uint64_t i = wave0; i < nent; i += nwave)
    { const uint8_t *src = text + line_start[6 * i];
      const uint64_t n   = hdr_pos[i + 1] - hdr_pos[i];            // includes the '\n'
      uint8_t       *dst = blob + hdr_pos[i];
      for (uint64_t k = lane; k + 1 < n; k += 64)
        dst[k] = src[k];
      if (lane == 0) dst[n - 1] = '\n';
    }
}

__global__ void k_add_one(uint32_t *v, uint64_t n)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i < n) v[i] += 1;
}

int dx_scan_u32(dx_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out /* n+1 */, uint64_t *total);
extern "C" int dx_parse_quiva_headers(const uint8_t *blob, const uint64_t *pos, uint64_t n, int32_t *hdr4,
                                      size_t *prefix_len, uint64_t *bad_entry);

extern "C" int dx_index_quiva_device(dx_ctx *ctx, const uint8_t *d_text, uint64_t nbytes,
                                     uint64_t **d_off, uint32_t **d_len, uint64_t *count,
                                     int32_t **hdr4, size_t *prefix_len, uint64_t *errline, int *errcode)
{ if (ctx == NULL) return DX_E_ARG;
  if (!d_off || !d_len || !count || !hdr4)
    return dx_fail(ctx, DX_E_ARG, "dx_index_quiva_device: NULL result pointer");
  *d_off = NULL; *d_len = NULL; *count = 0; *hdr4 = NULL;
  if (prefix_len) *prefix_len = 0;
  if (nbytes == 0) return DX_OK;
  if (d_text == NULL) return dx_fail(ctx, DX_E_ARG, "dx_index_quiva_device: NULL text");
  DX_HIP(ctx, hipSetDevice(ctx->device));

  const uint64_t ntiles = (nbytes + IDX_TILE - 1) / IDX_TILE;
  uint32_t *d_cnt = NULL, *d_hlen = NULL, *d_l = NULL;
  uint64_t *d_toff = NULL, *d_line = NULL, *d_hpos = NULL, *d_o = NULL, *pos = NULL;
  uint8_t  *d_blob = NULL, *blob = NULL, last = 0;
  uint64_t  nl = 0, nent = 0, hbytes = 0, bad = 0;
  unsigned long long *d_err = (unsigned long long *) (ctx->d_u64 + 8), err = ~0ull;
  int rc = DX_OK;
  #define CK(x)  do { rc = (x); if (rc != DX_OK) goto done; } while (0)
  #define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = dx_fail(ctx, DX_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
  #define FMT(line, code) do { if (errline) *errline = (line); if (errcode) *errcode = (code); rc = DX_E_FORMAT; goto done; } while (0)

  // ---- newline scan -> start of every line
  CKH(hipMalloc((void **) &d_cnt, ntiles * 4));
  CKH(hipMalloc((void **) &d_toff, (ntiles + 1) * 8));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_nl_count, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream, d_text, nbytes, d_cnt);
  dx_prof_end(ctx);
  CK(dx_scan_u32(ctx, d_cnt, ntiles, d_toff, &nl));
  CKH(hipMemcpyAsync(&last, d_text + nbytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  // The last line without a newline: a header or an entry's first line, "Last line does not end with a newline" (QV.c:771-781);
  // one of its lines 2-5, "not the same length" (QV.c:792: strlen, newline included, against the first line's) -- unless it is
  // ONE character longer than the others, which the reference takes: the host index (dx_index_quiva), which the file driver
  // asks after any refusal here, has that case.
  if (last != '\n') FMT(nl + 1, nl % 6 >= 2 ? DX_IDX_RAGGED : DX_IDX_NO_NEWLINE);
  CKH(hipMalloc((void **) &d_line, (nl + 2) * 8));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_nl_fill, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream,
                     d_text, nbytes, (const uint64_t *) d_toff, d_line);
  dx_prof_end(ctx);

  // ---- entries: structure checks + index
  nent = (nl + 5) / 6;                                              // a trailing partial entry is checked too
  if (nent == 0) goto done;
  CKH(hipMalloc((void **) &d_o, nent * 8 + 64));
  CKH(hipMalloc((void **) &d_l, nent * 4 + 64));
  CKH(hipMalloc((void **) &d_hlen, nent * 4));
  CKH(hipMemsetAsync(d_err, 0xff, 8, ctx->stream));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_qv_entries, dim3((unsigned) ((nent + DX_BLOCK - 1) / DX_BLOCK)), dim3(DX_BLOCK), 0, ctx->stream,
                     d_text, (const uint64_t *) d_line, nl, nent, d_o, d_l, d_hlen, d_err);
  dx_prof_end(ctx);
  CKH(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (err != ~0ull) FMT(err >> 8, (int) (err & 0xff));

  // ---- header lines to the host; sscanf there, exactly as QV.c:964
  hipLaunchKernelGGL(k_add_one, dim3((unsigned) ((nent + DX_BLOCK - 1) / DX_BLOCK)), dim3(DX_BLOCK), 0, ctx->stream, d_hlen, nent);
  CKH(hipMalloc((void **) &d_hpos, (nent + 1) * 8));
  CK(dx_scan_u32(ctx, d_hlen, nent, d_hpos, &hbytes));
  CKH(hipMalloc((void **) &d_blob, hbytes + 16));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_gather_headers, dim3((unsigned) dx_grid_waves(ctx, nent, 32)), dim3(DX_BLOCK), 0, ctx->stream,
                     d_text, (const uint64_t *) d_line, nent, (const uint64_t *) d_hpos, d_blob);
  dx_prof_end(ctx);
  blob  = (uint8_t *) malloc(hbytes + 16);
  pos   = (uint64_t *) malloc((nent + 1) * 8);
  *hdr4 = (int32_t *) malloc(nent * 4 * sizeof(int32_t));
  if (!blob || !pos || !*hdr4) { rc = DX_E_NOMEM; goto done; }
  CKH(hipMemcpyAsync(blob, d_blob, hbytes, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipMemcpyAsync(pos, d_hpos, (nent + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (dx_parse_quiva_headers(blob, pos, nent, *hdr4, prefix_len, &bad) != DX_OK)
    FMT(6 * bad + 1, DX_IDX_BAD_HEADER);
  *d_off = d_o; *d_len = d_l; *count = nent;
  d_o = NULL; d_l = NULL;

done:
  (void) hipStreamSynchronize(ctx->stream);
  (void) hipFree(d_cnt); (void) hipFree(d_toff); (void) hipFree(d_line); (void) hipFree(d_hlen);
  (void) hipFree(d_hpos); (void) hipFree(d_blob); (void) hipFree(d_o); (void) hipFree(d_l);
  free(blob); free(pos);
  if (rc != DX_OK && *hdr4) { free(*hdr4); *hdr4 = NULL; }
  #undef CK
  #undef CKH
  #undef FMT
  return rc;
}

// =============================================================================================
//  .fasta / .arrow: records are delimited by header lines (first byte '>'), dexta.c:139-183
// =============================================================================================

// per line: 1 if it is a header line (non-empty, starts with '>'), else 0; also the longest line
__global__ __launch_bounds__(DX_BLOCK)
void k_seq_lines(const uint8_t *text, const uint64_t *line_start, uint64_t nlines, uint32_t *is_hdr,
                 unsigned long long *err)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i >= nlines) return;
  const uint64_t s = line_start[i], e = line_start[i + 1] - 1;
  const bool h = e > s && text[s] == '>';
  is_hdr[i] = h ? 1u : 0u;
  if (e - s + 1 >= 100000ull)                                      // dexta.c:165-172: fgets limit
    atomicMin(err, ((unsigned long long) (i + 1) << 8) | DX_IDX_TOO_LONG);
  if (i == 0 && !h)
    atomicMin(err, (1ull << 8) | DX_IDX_NO_HEADER);                // dexta.c:113
}

// one thread per header line k-th record: needs rank of the header among headers
__global__ __launch_bounds__(DX_BLOCK)
void k_seq_records(const uint8_t *text, const uint64_t *line_start, uint64_t nlines, const uint32_t *is_hdr,
                   const uint64_t *hdr_rank /* exclusive scan of is_hdr, nlines+1 */, uint64_t nbytes,
                   uint64_t *hline /* line index of record r's header */)
{ const uint64_t i = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (i >= nlines || !is_hdr[i]) return;
  hline[hdr_rank[i]] = i;
  (void) text; (void) line_start; (void) nbytes;
}

__global__ __launch_bounds__(DX_BLOCK)
void k_seq_extent(const uint64_t *line_start, uint64_t nlines, const uint64_t *hline, uint64_t nrec,
                  uint64_t *off, uint32_t *tlen, uint32_t *nsym, uint32_t *hdr_len, unsigned long long *err)
{ const uint64_t r = (uint64_t) blockIdx.x * DX_BLOCK + threadIdx.x;
  if (r >= nrec) return;
  const uint64_t h  = hline[r];
  const uint64_t nx = r + 1 < nrec ? hline[r + 1] : nlines;        // first line after this record
  const uint64_t s  = line_start[h + 1], e = line_start[nx];
  const uint64_t tl = e - s, lines = nx - h - 1;
  off[r] = s;
  hdr_len[r] = (uint32_t) (line_start[h + 1] - 1 - line_start[h]);
  if (tl - lines > 0x7fffffffull)
    { atomicMin(err, ((unsigned long long) nx << 8) | DX_IDX_TOO_LONG);
      tlen[r] = nsym[r] = 0;
      return;
    }
  tlen[r] = (uint32_t) tl;
  nsym[r] = (uint32_t) (tl - lines);
}

__global__ __launch_bounds__(DX_BLOCK)
void k_gather_lines(const uint8_t *text, const uint64_t *line_start, const uint64_t *hline, uint64_t nrec,
                    const uint64_t *hdr_pos, uint8_t *blob)
{ const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  for (uint64_t i = wave0; i < nrec; i += nwave)
    { const uint8_t *src = text + line_start[hline[i]];
      const uint64_t n   = hdr_pos[i + 1] - hdr_pos[i];
      uint8_t       *dst = blob + hdr_pos[i];
      for (uint64_t k = lane; k + 1 < n; k += 64)
        dst[k] = src[k];
      if (lane == 0) dst[n - 1] = '\n';
    }
}

extern "C" int dx_parse_seq_headers(int arrow, const uint8_t *blob, const uint64_t *pos, uint64_t n,
                                    int32_t *hdr4, uint16_t *cnr4, size_t *prefix_len, uint64_t *bad_entry);

extern "C" int dx_index_seq_device(dx_ctx *ctx, int arrow, const uint8_t *d_text, uint64_t nbytes,
                                   uint64_t **d_off, uint32_t **d_tlen, uint32_t **d_nsym, uint64_t *count,
                                   int32_t **hdr4, uint16_t **cnr4, size_t *prefix_len,
                                   uint64_t *errline, int *errcode)
{ if (ctx == NULL) return DX_E_ARG;
  if (!d_off || !d_tlen || !d_nsym || !count || !hdr4 || !cnr4)
    return dx_fail(ctx, DX_E_ARG, "dx_index_seq_device: NULL result pointer");
  *d_off = NULL; *d_tlen = NULL; *d_nsym = NULL; *count = 0; *hdr4 = NULL; *cnr4 = NULL;
  if (prefix_len) *prefix_len = 0;
  if (nbytes == 0)
    { if (errline) *errline = 1;
      if (errcode) *errcode = DX_IDX_EMPTY;
      return DX_E_FORMAT;
    }
  if (d_text == NULL) return dx_fail(ctx, DX_E_ARG, "dx_index_seq_device: NULL text");
  DX_HIP(ctx, hipSetDevice(ctx->device));

  const uint64_t ntiles = (nbytes + IDX_TILE - 1) / IDX_TILE;
  uint32_t *d_cnt = NULL, *d_ish = NULL, *d_hlen = NULL, *d_tl = NULL, *d_ns = NULL;
  uint64_t *d_toff = NULL, *d_line = NULL, *d_rank = NULL, *d_hline = NULL, *d_hpos = NULL, *d_o = NULL, *pos = NULL;
  uint8_t  *d_blob = NULL, *blob = NULL, last = 0;
  uint64_t  nl = 0, nrec = 0, hbytes = 0, bad = 0;
  unsigned long long *d_err = (unsigned long long *) (ctx->d_u64 + 8), err = ~0ull;
  int rc = DX_OK;
  #define CK(x)  do { rc = (x); if (rc != DX_OK) goto done; } while (0)
  #define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = dx_fail(ctx, DX_E_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
  #define FMT(line, code) do { if (errline) *errline = (line); if (errcode) *errcode = (code); rc = DX_E_FORMAT; goto done; } while (0)
  #define GRID(n_) dim3((unsigned) (((n_) + DX_BLOCK - 1) / DX_BLOCK))

  CKH(hipMalloc((void **) &d_cnt, ntiles * 4));
  CKH(hipMalloc((void **) &d_toff, (ntiles + 1) * 8));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_nl_count, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream, d_text, nbytes, d_cnt);
  dx_prof_end(ctx);
  CK(dx_scan_u32(ctx, d_cnt, ntiles, d_toff, &nl));
  CKH(hipMemcpyAsync(&last, d_text + nbytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (last != '\n') FMT(nl + 1, DX_IDX_TOO_LONG);                   // dexta.c:165-172 (no newline = "too long")
  CKH(hipMalloc((void **) &d_line, (nl + 2) * 8));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_nl_fill, dim3((unsigned) ntiles), dim3(DX_BLOCK), 0, ctx->stream,
                     d_text, nbytes, (const uint64_t *) d_toff, d_line);
  dx_prof_end(ctx);

  CKH(hipMalloc((void **) &d_ish, nl * 4 + 4));
  CKH(hipMalloc((void **) &d_rank, (nl + 1) * 8));
  CKH(hipMemsetAsync(d_err, 0xff, 8, ctx->stream));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_seq_lines, GRID(nl), dim3(DX_BLOCK), 0, ctx->stream, d_text, (const uint64_t *) d_line, nl, d_ish, d_err);
  dx_prof_end(ctx);
  CK(dx_scan_u32(ctx, d_ish, nl, d_rank, &nrec));
  CKH(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (err != ~0ull) FMT(err >> 8, (int) (err & 0xff));
  if (nrec == 0) FMT(1, DX_IDX_NO_HEADER);

  CKH(hipMalloc((void **) &d_hline, nrec * 8));
  CKH(hipMalloc((void **) &d_o, nrec * 8 + 64));
  CKH(hipMalloc((void **) &d_tl, nrec * 4 + 64));
  CKH(hipMalloc((void **) &d_ns, nrec * 4 + 64));
  CKH(hipMalloc((void **) &d_hlen, nrec * 4));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_seq_records, GRID(nl), dim3(DX_BLOCK), 0, ctx->stream, d_text, (const uint64_t *) d_line, nl,
                     (const uint32_t *) d_ish, (const uint64_t *) d_rank, nbytes, d_hline);
  hipLaunchKernelGGL(k_seq_extent, GRID(nrec), dim3(DX_BLOCK), 0, ctx->stream, (const uint64_t *) d_line, nl,
                     (const uint64_t *) d_hline, nrec, d_o, d_tl, d_ns, d_hlen, d_err);
  dx_prof_end(ctx);
  CKH(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (err != ~0ull) FMT(err >> 8, (int) (err & 0xff));

  hipLaunchKernelGGL(k_add_one, GRID(nrec), dim3(DX_BLOCK), 0, ctx->stream, d_hlen, nrec);
  CKH(hipMalloc((void **) &d_hpos, (nrec + 1) * 8));
  CK(dx_scan_u32(ctx, d_hlen, nrec, d_hpos, &hbytes));
  CKH(hipMalloc((void **) &d_blob, hbytes + 16));
  dx_prof_begin(ctx, DX_K_INDEX);
  hipLaunchKernelGGL(k_gather_lines, dim3((unsigned) dx_grid_waves(ctx, nrec, 32)), dim3(DX_BLOCK), 0, ctx->stream,
                     d_text, (const uint64_t *) d_line, (const uint64_t *) d_hline, nrec, (const uint64_t *) d_hpos, d_blob);
  dx_prof_end(ctx);
  blob  = (uint8_t *) malloc(hbytes + 16);
  pos   = (uint64_t *) malloc((nrec + 1) * 8);
  *hdr4 = (int32_t *) malloc(nrec * 4 * sizeof(int32_t));
  *cnr4 = (uint16_t *) malloc(nrec * 4 * sizeof(uint16_t));
  if (!blob || !pos || !*hdr4 || !*cnr4) { rc = DX_E_NOMEM; goto done; }
  CKH(hipMemcpyAsync(blob, d_blob, hbytes, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipMemcpyAsync(pos, d_hpos, (nrec + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  CKH(hipStreamSynchronize(ctx->stream));
  if (dx_parse_seq_headers(arrow, blob, pos, nrec, *hdr4, *cnr4, prefix_len, &bad) != DX_OK)
    FMT(0, DX_IDX_BAD_HEADER);                                      // exact line: the host indexer reports it
  *d_off = d_o; *d_tlen = d_tl; *d_nsym = d_ns; *count = nrec;
  d_o = NULL; d_tl = NULL; d_ns = NULL;

done:
  (void) hipStreamSynchronize(ctx->stream);
  (void) hipFree(d_cnt); (void) hipFree(d_toff); (void) hipFree(d_line); (void) hipFree(d_ish); (void) hipFree(d_rank);
  (void) hipFree(d_hline); (void) hipFree(d_hlen); (void) hipFree(d_hpos); (void) hipFree(d_blob);
  (void) hipFree(d_o); (void) hipFree(d_tl); (void) hipFree(d_ns);
  free(blob); free(pos);
  if (rc != DX_OK)
    { if (*hdr4) { free(*hdr4); *hdr4 = NULL; }
      if (*cnr4) { free(*cnr4); *cnr4 = NULL; }
    }
  #undef CK
  #undef CKH
  #undef FMT
  #undef GRID
  return rc;
}
