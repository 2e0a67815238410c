/* dx_layout.h -- layout of the group index (dx_qv_subindex): the ONE definition, shared by the device code that writes
 * and reads it (dx_device.hpp includes this file; the functions are __host__ __device__ there) and the host walk that can
 * produce it for a bare file (plain C).                                                                            */
#ifndef DX_LAYOUT_H
#define DX_LAYOUT_H
#include <stdint.h>
#ifdef __HIPCC__
#define DXL_FN __host__ __device__ __forceinline__
#else
#define DXL_FN static inline
#endif

#define DXL_SUB_NONE     255u                 /* first byte of a plain line's share: no index for it */
#define DXL_RUN_NONE     0xffffffffu          /* header word of a run-coded line: no index for it */
#define DXL_RUN_EIGHTS   0x80000000u          /* in the third header word (the deletion line's passes): the LAST pass's tokens are dealt 8 a
                                                 lane like every other pass's (the device walk's index), not (m + 63) / 64 a lane (the encoder's) */
#define DXL_SYNC_NONE    0xffffffffu          /* first word of a plain line's share in the device walk's index (its other words: where symbol 64 g
                                                 is, k_qv_decode_sync): no words for this line */
#define DXL_RUN_PASSBITS 13312u               /* bits the 64 groups of a pass may take together */
#define DXL_RUN_PASS     512u                 /* tokens of a pass: 64 groups of up to 8 */
#define DXL_RUN_STRETCH  5120u                /* positions of a pass the decoder stages in LDS at a time */

DXL_FN uint32_t dxl_sub_groups(uint32_t L) { return (L + 15u) >> 4; }
DXL_FN uint32_t dxl_sub_words(uint32_t L)  { return (dxl_sub_groups(L) + 3u) >> 2; }
DXL_FN uint32_t dxl_run_passes(uint32_t tokens) { return (tokens + 511u) >> 9; }
DXL_FN uint32_t dxl_run_base(uint32_t L)   { return 4u * dxl_sub_words(L); }
/* words of an entry: the four plain shares, three header words, 64 group words per pass of either run-coded line */
DXL_FN uint64_t dxl_entry_words(uint32_t L, uint32_t passes_del, uint32_t passes_sub)
{ return (uint64_t) dxl_run_base(L) + 3u + 64ull * ((uint64_t) passes_del + passes_sub); }
/* tokens a run-coded line can have at most (one per symbol; the slots of k_qv_hist's token hand-over are sized per batch) */
DXL_FN uint32_t dxl_tok_limit(uint32_t L) { return L + 8u; }
#endif
