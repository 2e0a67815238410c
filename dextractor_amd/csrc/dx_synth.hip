// dx_synth.hip -- seeded synthetic .quiva corpus written directly into HBM (benchmark plumbing).
//
// Mirrors dextractor_amd/synth.py bit for bit: every symbol is lut[stream][hash(seed, entry,
// stream, position) >> 20] with the 32-bit counter hash below, the tag is 'N' exactly where the
// deletion QV equals the run character.  One wavefront per entry, 16 bytes per lane per step.
#include "dx_internal.hpp"
#include "dx_device.hpp"

__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x)
{ x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ uint32_t stream_key(uint32_t seed, uint64_t entry, uint32_t stream)
{ const uint32_t lo = (uint32_t) entry, hi = (uint32_t) (entry >> 32);
  const uint32_t k0 = lowbias32(seed + 0x9e3779b9u * (lo + 1u));
  return lowbias32(k0 ^ (hi * 0x85ebca6bu + (stream * 0xc2b2ae35u + 0x27d4eb2fu)));
}

__device__ __forceinline__ uint32_t sample12(uint32_t key, uint32_t pos)
{ return lowbias32(key + pos * 0x9e3779b1u) >> 20; }

struct synth_args
{ uint32_t        seed;
  uint64_t        entry0, n;
  const uint64_t *off;
  const uint32_t *len;
  const int32_t  *hdr4;
  const uint8_t  *lut;          // 5 x 4096: del, tag, ins, mrg, sub
  int             del_run;
  char            movie[64];
  uint32_t        mlen;
  uint8_t        *text;
};

__device__ __forceinline__ void put_dec(uint8_t *p, uint32_t v, int digits)
{ for (int k = digits - 1; k >= 0; k--)
    { p[k] = (uint8_t) ('0' + v % 10u);
      v /= 10u;
    }
}

__global__ __launch_bounds__(DX_BLOCK)
void k_synth_quiva(synth_args a)
{ __shared__ uint8_t s_lut[5 * 4096];
  for (int k = threadIdx.x; k < 5 * 4096 / 4; k += DX_BLOCK)
    ((uint32_t *) s_lut)[k] = ((const uint32_t *) a.lut)[k];
  __syncthreads();

  const int      lane  = lane_id();
  const uint64_t wave0 = (uint64_t) blockIdx.x * DX_WAVES_PER_BLK + (threadIdx.x >> 6);
  const uint64_t nwave = (uint64_t) gridDim.x * DX_WAVES_PER_BLK;
  static const int lut_row[5]   = { 0, 1, 2, 3, 4 };
  static const int stream_id[5] = { 0, 1, 2, 3, 4 };      // S_DEL, S_TAG, S_INS, S_MRG, S_SUB

  for (uint64_t r = wave0; r < a.n; r += nwave)
    { const uint32_t L    = a.len[r];
      uint8_t       *body = a.text + a.off[r];

      if (lane == 0)      // "@<movie>/WWWWWWWW/BBBBBBB_EEEEEEE RQ=0.QQQ\n" ending right before the body
        { const int32_t *h  = a.hdr4 + 4 * r;
          const uint32_t hl = 1 + a.mlen + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1;
          uint8_t *p = body - hl;
          *p++ = '@';
          for (uint32_t k = 0; k < a.mlen; k++) *p++ = (uint8_t) a.movie[k];
          *p++ = '/';  put_dec(p, (uint32_t) h[0], 8); p += 8;
          *p++ = '/';  put_dec(p, (uint32_t) h[1], 7); p += 7;
          *p++ = '_';  put_dec(p, (uint32_t) h[2], 7); p += 7;
          *p++ = ' '; *p++ = 'R'; *p++ = 'Q'; *p++ = '='; *p++ = '0'; *p++ = '.';
          put_dec(p, (uint32_t) h[3], 3); p += 3;
          *p = '\n';
        }

      const uint32_t kdel = stream_key(a.seed, a.entry0 + r, 0);
      for (int s = 0; s < 5; s++)
        { const uint32_t key  = stream_key(a.seed, a.entry0 + r, (uint32_t) stream_id[s]);
          const uint8_t *lut  = s_lut + 4096 * lut_row[s];
          uint8_t       *line = body + (uint64_t) s * ((uint64_t) L + 1u);
          for (uint32_t base = 0; base < L; base += DX_STEP)
            { const uint32_t pos = base + 16u * lane;
              if (pos >= L) continue;
              const int valid = L - pos >= 16u ? 16 : (int) (L - pos);
              uint32_t w[4] = { 0u, 0u, 0u, 0u };
              #pragma unroll
              for (int b = 0; b < 16; b++)
                { uint32_t ch = lut[sample12(key, pos + b)];
                  if (s == 1 && a.del_run >= 0 && s_lut[sample12(kdel, pos + b)] == (uint32_t) a.del_run)
                    ch = 'N';
                  w[b >> 2] |= ch << (8 * (b & 3));
                }
              if (valid == 16)
                { u32x4 v = { w[0], w[1], w[2], w[3] };
                  *(u32x4_u *) (line + pos) = v;
                }
              else
                for (int b = 0; b < valid; b++)
                  line[pos + b] = (uint8_t) (w[b >> 2] >> (8 * (b & 3)));
            }
          if (lane == 0)
            line[L] = '\n';
        }
    }
}

extern "C" int dx_synth_quiva(dx_ctx *ctx, uint32_t seed, uint64_t entry0, uint64_t n,
                              const uint64_t *d_off, const uint32_t *d_len, const int32_t *d_hdr4,
                              const uint8_t *d_lut, int del_run, const char *movie, uint8_t *d_text)
{ if (ctx == NULL) return DX_E_ARG;
  if (n == 0) return DX_OK;
  if (!d_off || !d_len || !d_hdr4 || !d_lut || !d_text || !movie)
    return dx_fail(ctx, DX_E_ARG, "dx_synth_quiva: NULL argument");
  synth_args a;
  a.seed = seed; a.entry0 = entry0; a.n = n; a.off = d_off; a.len = d_len; a.hdr4 = d_hdr4;
  a.lut = d_lut; a.del_run = del_run; a.text = d_text;
  a.mlen = (uint32_t) strlen(movie);
  if (a.mlen >= sizeof(a.movie))
    return dx_fail(ctx, DX_E_ARG, "dx_synth_quiva: movie name too long");
  memcpy(a.movie, movie, a.mlen + 1);
  DX_HIP(ctx, hipSetDevice(ctx->device));
  DX_LAUNCH(ctx, DX_K_SYNTH, k_synth_quiva, dx_grid_waves(ctx, n, 16), DX_BLOCK, a);
  return DX_OK;
}
