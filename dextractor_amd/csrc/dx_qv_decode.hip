// dx_qv_decode.hip -- Uncompress_Next_QVentry (QV.c:1428-1481) on the device.  (round-1 stub)
#include "dx_internal.hpp"

extern "C" int dx_qv_decode(dx_ctx *ctx, const uint8_t *d_in, const uint64_t *d_seg_off, const uint32_t *d_len,
                            uint64_t n, int upper, uint8_t *d_out, const uint64_t *d_out_off)
{ (void) d_in; (void) d_seg_off; (void) d_len; (void) n; (void) upper; (void) d_out; (void) d_out_off;
  if (ctx == NULL) return DX_E_ARG;
  return dx_fail(ctx, DX_E_UNSUPPORTED, "dx_qv_decode: not implemented yet");
}
