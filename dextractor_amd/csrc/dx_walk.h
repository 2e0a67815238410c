/*
 * dx_walk.h -- what the host (dx_host.c: the look-up tables of the record walk) and the device walk (dx_qv_walk.hip) share.
 *
 * A bare .dexqv stores no record or segment lengths (QV.c:1428-1481, undexqv.c:119-208): a segment's end is known only
 * after every code of it has been passed.  The tables below are the host walk's (dx_host.c: wlut, mwlut, rwlut), packed
 * to 16 bits an entry so that the six a walk needs fit a workgroup's LDS twice over per CU.
 *
 * One blob, WALK_BLOB_BYTES long, little endian:
 *   w16  [6][65536] u16   by 16-bit window: len << 8 | symbol (0: no such code), the schemes DX_DEL .. DX_SRUN   (stays in
 *                         memory: codes of more than 12 bits, escape literals)
 *   mw   [4][4096]  u16   by 12-bit window, the symbol schemes DX_DEL .. DX_SUB: bits of the whole codes the window holds |
 *                         the last one's length << 4 | their number << 8; 0: none whole, or an escape code first (the 8-bit
 *                         literal behind it is no code)
 *   rw   [2][4096]  u16   by 12-bit window, del and sub: a (run code, symbol code) pair whole in the window, no literal:
 *                         bits | symbol code's length << 4 | (run + 1) << 8; 0: no such pair (runs of 255 and more have
 *                         literals: run + 1 <= 255)
 *   r1   [2][4096]  u16   by 12-bit window, the run schemes: the run code alone, in the same form: its bits | its bits << 4 |
 *                         the run << 8; 0: longer than the window, no such code, or the code of 255 (a 16-bit literal follows)
 *   one  [4][4096]  u16   by 12-bit window, the symbol schemes: the FIRST code alone, in the same form: its bits | its bits << 4 |
 *                         1 << 8; 0: longer than the window, an escape code, none
 * All four kinds read alike -- bits to skip, the last code's length, the symbols covered -- so one look-up loop serves every
 * line and every kind of step.
 */
#ifndef DX_WALK_H
#define DX_WALK_H

#include <stdint.h>
#include <stddef.h>
#include "dexgpu.h"

#define WALK_WIN        12
#define WALK_W16_OFF    0u
#define WALK_MW_OFF     (6u * 65536u * 2u)
#define WALK_RW_OFF     (WALK_MW_OFF + 4u * 4096u * 2u)
#define WALK_R1_OFF     (WALK_RW_OFF + 2u * 4096u * 2u)
#define WALK_ONE_OFF    (WALK_R1_OFF + 2u * 4096u * 2u)
#define WALK_BLOB_BYTES (WALK_ONE_OFF + 4u * 4096u * 2u)

#ifdef __cplusplus
extern "C" {
#endif

/* fills blob (WALK_BLOB_BYTES) from a coding; esc[k] = scheme k (DX_DEL .. DX_SUB) is of the escape kind */
int dx_walk_luts_build(const dx_qv_coding *cd, uint8_t *blob, int esc[4]);

#ifdef __cplusplus
}
#endif
#endif
