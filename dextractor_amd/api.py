"""Python mirror of the libdexgpu C-ABI (thin: device buffers, numpy in/out, errors -> exceptions).

Names follow the reference's tools and functions: `dexta/undexta/dexar/undexar/dexqv` are the
whole-file drivers (dexta.c ... dexqv.c); `Context.qv_*` are the batch forms of QVcoding_Scan,
Create_QVcoding and Compress_Next_QVentry (QV.h:48-97).  All compute happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class DevView:
    """A pointer into a DevBuf (which stays the owner)."""

    def __init__(self, base: "DevBuf", byte_offset: int):
        self.base, self.ptr = base, base.ptr + int(byte_offset)


class DevBuf:
    """A device allocation owned by a Context."""

    def offset(self, nbytes: int) -> DevView:
        return DevView(self, nbytes)

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._chk(ctx.lib.dx_malloc(ctx.h, self.nbytes + 64, C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.ctx.lib.dx_free(self.ctx.h, self.ptr)
            self.ptr = None

    def upload(self, arr) -> "DevBuf":
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        if a.nbytes:
            self.ctx._chk(self.ctx.lib.dx_h2d(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype=np.uint8, count=None, offset=0) -> np.ndarray:
        dt = np.dtype(dtype)
        count = (self.nbytes - offset) // dt.itemsize if count is None else count
        out = np.empty(count, dtype=dt)
        if out.nbytes:
            self.ctx._chk(self.ctx.lib.dx_d2h(self.ctx.h, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        self.ctx._chk(self.ctx.lib.dx_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self


class DeviceIndex:
    """The arrays dx_qv_walk_device left on the device (the library's memory: free())."""

    class _View:
        def __init__(self, ptr): self.ptr = ptr

    def __init__(self, ctx, x):
        self.ctx, self.x, self.n = ctx, x, int(x.n)
        self.pieces, self.piece_bytes = int(x.pieces), int(x.piece_bytes)
        self.rec_off, self.hdr_off = self._View(x.d_rec_off), self._View(x.d_hdr_off)
        self.seg, self.len, self.hdr4 = self._View(x.d_seg), self._View(x.d_len), self._View(x.d_hdr4)
        self.gidx_words, self.gidx_none, self.gidx_nosync = int(x.gidx_words), int(x.gidx_none), int(x.gidx_nosync)
        self.gidx = self._View(x.d_gidx) if x.d_gidx else None
        self.gidx_off = self._View(x.d_gidx_off) if x.d_gidx_off else None

    def use(self, d_in):
        """dx_qv_use_dindex: the run-coded lines' groups this walk has noted go to the decoder (d_in: the stream's device
        address, as dx_qv_decode will get it); use(None) takes them back."""
        self.ctx._chk(self.ctx.lib.dx_qv_use_dindex(self.ctx.h, d_in.ptr if d_in is not None else None,
                                                    C.byref(self.x) if d_in is not None else None))

    def download(self):
        """-> dict of numpy arrays, as qv_walk returns them"""
        n, out = self.n, {}
        for name, v, dt, shape in (("rec_off", self.rec_off, np.uint64, (n + 1,)), ("hdr_off", self.hdr_off, np.uint64, (n + 1,)),
                                   ("seg", self.seg, np.uint32, (n, 5)), ("len", self.len, np.uint32, (n,)), ("hdr4", self.hdr4, np.int32, (n, 4))):
            a = np.zeros(shape, dt)
            if a.nbytes:
                self.ctx._chk(self.ctx.lib.dx_d2h(self.ctx.h, a.ctypes.data, v.ptr, a.nbytes))
            out[name] = a
        out["n"] = n
        return out

    def free(self):
        if self.x is not None:
            self.ctx.lib.dx_qv_dindex_free(self.ctx.h, C.byref(self.x))
            self.x = None


class Context:
    def __init__(self, device: int = 0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.dx_open(device, C.byref(h))
        if rc != 0:
            raise L.DexGPUError(rc, (self.lib.dx_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.dx_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise L.DexGPUError(rc, (self.lib.dx_last_error(self.h) or b"").decode())

    # ---- memory ------------------------------------------------------------------------------
    def alloc(self, nbytes) -> DevBuf:
        return DevBuf(self, nbytes)

    def to_device(self, arr) -> DevBuf:
        a = np.ascontiguousarray(arr)
        return DevBuf(self, a.nbytes).upload(a)

    def sync(self):
        self._chk(self.lib.dx_sync(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.dx_set_stream(self.h, stream_ptr))

    # ---- profiling ---------------------------------------------------------------------------
    def profile(self, enable=True):
        self._chk(self.lib.dx_profile(self.h, int(enable)))

    def kernel_times(self) -> dict:
        out = {}
        for k, name in enumerate(L.KERNELS):
            ms, cnt = C.c_double(), C.c_uint64()
            self._chk(self.lib.dx_profile_get(self.h, k, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[name] = (ms.value, cnt.value)
        return out

    # ---- 2-bit packers -----------------------------------------------------------------------
    def pack2_encode(self, alphabet, d_text, d_off, d_tlen, d_nsym, n, d_hdr, d_hdr_off, d_out, d_out_off):
        self._chk(self.lib.dx_pack2_encode(self.h, alphabet, d_text.ptr, d_off.ptr, d_tlen.ptr, d_nsym.ptr, n,
                                           d_hdr.ptr if d_hdr else None, d_hdr_off.ptr if d_hdr_off else None,
                                           d_out.ptr, d_out_off.ptr))

    def pack2_decode(self, letters, d_in, d_in_off, d_nsym, n, width, d_out, d_out_off):
        self._chk(self.lib.dx_pack2_decode(self.h, letters, d_in.ptr, d_in_off.ptr, d_nsym.ptr, n, width,
                                           d_out.ptr, d_out_off.ptr))

    # ---- QV coder ----------------------------------------------------------------------------
    @staticmethod
    def qv_batch(d_text, d_off, d_len, n, line_pad=1, text_bytes=0) -> L.QVBatch:
        return L.QVBatch(d_text.ptr, d_off.ptr, d_len.ptr, n, text_bytes, line_pad)

    def qv_lossy_text(self, batch):
        """QV.c:1355-1372's rounding of the insertion / merge QVs, in place on the batch's device text."""
        self._chk(self.lib.dx_qv_lossy_text(self.h, C.byref(batch)))

    def qv_prescan(self, batch, entry0=0, params=None) -> L.QVParams:
        p = params or L.QVParams(-1, -1, -1, -1)
        self._chk(self.lib.dx_qv_prescan(self.h, C.byref(batch), entry0, C.byref(p)))
        return p

    def qv_hist(self, batch, params, entry0=0, hist=None, tot=0):
        """Returns (hist uint64 [6,256], totChar); adds into `hist`/`tot` when given (shards)."""
        h = L.HIST()
        if hist is not None:
            np.ctypeslib.as_array(h)[:] = hist
        t = C.c_uint64(tot)
        self._chk(self.lib.dx_qv_hist(self.h, C.byref(batch), entry0, C.byref(params), C.byref(h), C.byref(t)))
        return np.ctypeslib.as_array(h).reshape(6, 256).copy(), t.value

    def qv_scan(self, batch, entry0=0, params=None, hist=None, tot=0):
        """dx_qv_scan: qv_prescan + qv_hist with one wait on the device.  Returns (params, hist uint64 [6,256], totChar)."""
        p = params or L.QVParams(-1, -1, -1, -1)
        h = L.HIST()
        if hist is not None:
            np.ctypeslib.as_array(h)[:] = hist
        t = C.c_uint64(tot)
        self._chk(self.lib.dx_qv_scan(self.h, C.byref(batch), entry0, C.byref(p), C.byref(h), C.byref(t)))
        return p, np.ctypeslib.as_array(h).reshape(6, 256).copy(), t.value

    def qv_set_coding(self, coding, lossy=False):
        self._chk(self.lib.dx_qv_set_coding(self.h, C.byref(coding), int(lossy)))

    def qv_sizes(self, batch, d_hdr_off, d_seg, d_rec_off) -> int:
        tot = C.c_uint64()
        self._chk(self.lib.dx_qv_sizes(self.h, C.byref(batch), d_hdr_off.ptr if d_hdr_off else None,
                                       d_seg.ptr, d_rec_off.ptr, C.byref(tot)))
        return tot.value

    def qv_encode(self, batch, d_hdr, d_hdr_off, d_rec_off, d_seg, d_out):
        self._chk(self.lib.dx_qv_encode(self.h, C.byref(batch), d_hdr.ptr if d_hdr else None,
                                        d_hdr_off.ptr if d_hdr_off else None, d_rec_off.ptr, d_seg.ptr,
                                        d_out.ptr))

    def qv_encode_onepass(self, batch, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap) -> int:
        """dx_qv_encode_onepass: sizes, offsets and the record stream without a size pass; returns the total."""
        tot = C.c_uint64()
        self._chk(self.lib.dx_qv_encode_onepass(self.h, C.byref(batch), d_hdr.ptr if d_hdr else None,
                                                d_hdr_off.ptr if d_hdr_off else None, d_seg.ptr, d_rec_off.ptr,
                                                d_out.ptr, out_cap, C.byref(tot)))
        return tot.value

    def qv_encode_onepass_begin(self, batch, d_hdr, d_hdr_off, d_seg, d_rec_off, d_out, out_cap):
        """dx_qv_encode_onepass_begin: queue the encode; the next batch's prescan / hist / build / set_coding may follow."""
        self._chk(self.lib.dx_qv_encode_onepass_begin(self.h, C.byref(batch), d_hdr.ptr if d_hdr else None,
                                                      d_hdr_off.ptr if d_hdr_off else None, d_seg.ptr, d_rec_off.ptr,
                                                      d_out.ptr, out_cap))

    def qv_encode_onepass_end(self) -> int:
        """dx_qv_encode_onepass_end: wait for the encode that has begun; returns the stream's size."""
        tot = C.c_uint64()
        self._chk(self.lib.dx_qv_encode_onepass_end(self.h, C.byref(tot)))
        return tot.value

    def qv_decode(self, d_in, d_rec_off, d_hdr_off, d_seg, d_len, n, upper, d_out, d_out_off, flip=False):
        flags = (1 if upper else 0) | (2 if flip else 0)          # DX_DECODE_UPPER | DX_DECODE_FLIP
        self._chk(self.lib.dx_qv_decode(self.h, d_in.ptr, d_rec_off.ptr, d_hdr_off.ptr if d_hdr_off else None,
                                        d_seg.ptr, d_len.ptr, n, flags, d_out.ptr, d_out_off.ptr))

    def qv_walk_device(self, d_img, nbytes, first, coding, newv=1, flip=0):
        """dx_qv_walk_device: the record walk of the bare stream at d_img on the device -> DeviceIndex (device arrays
        rec_off, hdr_off, seg, len, hdr4 as views that dx_qv_decode takes; .free() when done).  Raises DexGPUError
        (DX_E_MISMATCH) when the stream has to be walked on the host."""
        x = L.QVDIndex()
        self._chk(self.lib.dx_qv_walk_device(self.h, d_img.ptr, nbytes, first, C.byref(coding), newv, flip, C.byref(x)))
        return DeviceIndex(self, x)

    def index_quiva_device(self, d_text, nbytes):
        """GPU text front end -> (off uint64, len uint32, hdr4 int32 [n,4], prefix_len); raises
        DexGPUError(DX_E_FORMAT) with .line / .idx_code on a malformed image."""
        d_off, d_len, hdr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cnt, pl, line, code = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
        rc = self.lib.dx_index_quiva_device(self.h, d_text.ptr, nbytes, C.byref(d_off), C.byref(d_len), C.byref(cnt),
                                            C.byref(hdr), C.byref(pl), C.byref(line), C.byref(code))
        if rc != 0:
            e = L.DexGPUError(rc, f"line {line.value}: DX_IDX code {code.value}")
            e.line, e.idx_code = line.value, code.value
            raise e
        n = cnt.value
        off, ln = np.empty(n, np.uint64), np.empty(n, np.uint32)
        if n:
            self._chk(self.lib.dx_d2h(self.h, off.ctypes.data, d_off, n * 8))
            self._chk(self.lib.dx_d2h(self.h, ln.ctypes.data, d_len, n * 4))
            h4 = np.ctypeslib.as_array(C.cast(hdr, C.POINTER(C.c_int32)), (n, 4)).copy()
            self.lib.dx_free(self.h, d_off); self.lib.dx_free(self.h, d_len)
            C.CDLL(None).free(hdr)
        else:
            h4 = np.empty((0, 4), np.int32)
        return off, ln, h4, pl.value

    def index_seq_device(self, d_text, nbytes, arrow=False):
        """GPU text front end for .fasta/.arrow -> (off, tlen, nsym, hdr4, cnr4, prefix_len)."""
        d_off, d_tl, d_ns, hdr, cnr = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cnt, pl, line, code = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
        rc = self.lib.dx_index_seq_device(self.h, int(arrow), d_text.ptr, nbytes, C.byref(d_off), C.byref(d_tl),
                                          C.byref(d_ns), C.byref(cnt), C.byref(hdr), C.byref(cnr), C.byref(pl),
                                          C.byref(line), C.byref(code))
        if rc != 0:
            e = L.DexGPUError(rc, f"line {line.value}: DX_IDX code {code.value}")
            e.line, e.idx_code = line.value, code.value
            raise e
        n = cnt.value
        off, tl, ns = np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty(n, np.uint32)
        self._chk(self.lib.dx_d2h(self.h, off.ctypes.data, d_off, n * 8))
        self._chk(self.lib.dx_d2h(self.h, tl.ctypes.data, d_tl, n * 4))
        self._chk(self.lib.dx_d2h(self.h, ns.ctypes.data, d_ns, n * 4))
        h4 = np.ctypeslib.as_array(C.cast(hdr, C.POINTER(C.c_int32)), (n, 4)).copy()
        c4 = np.ctypeslib.as_array(C.cast(cnr, C.POINTER(C.c_uint16)), (n, 4)).copy()
        for p_ in (d_off, d_tl, d_ns):
            self.lib.dx_free(self.h, p_)
        C.CDLL(None).free(hdr); C.CDLL(None).free(cnr)
        return off, tl, ns, h4, c4, pl.value

    def synth_quiva(self, seed, entry0, n, d_off, d_len, d_hdr4, d_lut, del_run, movie, d_text):
        self._chk(self.lib.dx_synth_quiva(self.h, seed & 0xFFFFFFFF, entry0, n, d_off.ptr, d_len.ptr, d_hdr4.ptr,
                                          d_lut.ptr, del_run, movie.encode(), d_text.ptr))

    # ---- whole-file drivers (what the CLI tools call) -------------------------------------------
    def _file_call(self, fn, *args):
        out, n = C.c_void_p(), C.c_size_t()
        line, code = C.c_uint64(), C.c_int()
        extra = () if fn in (self.lib.dx_file_unpack2, self.lib.dx_file_undexqv) else (C.byref(line), C.byref(code))
        rc = fn(self.h, *args, C.byref(out), C.byref(n), *extra)
        if rc != 0:
            msg = (self.lib.dx_last_error(self.h) or b"").decode()
            if rc == -3 and extra:
                msg = f"line {line.value}: input rejected (DX_IDX code {code.value}) {msg}"
            raise L.DexGPUError(rc, msg)
        try:
            return C.string_at(out.value, n.value)
        finally:
            self.lib.dx_file_free(out)

    def dexta(self, fasta: bytes) -> bytes:
        return self._file_call(self.lib.dx_file_pack2, 0, fasta, len(fasta))

    def dexar(self, arrow: bytes) -> bytes:
        return self._file_call(self.lib.dx_file_pack2, 1, arrow, len(arrow))

    def undexta(self, img: bytes, upper=False, width=80) -> bytes:
        return self._file_call(self.lib.dx_file_unpack2, L.DX_LETTERS_UPPER if upper else L.DX_LETTERS_LOWER,
                               img, len(img), width)

    def undexar(self, img: bytes, width=80) -> bytes:
        return self._file_call(self.lib.dx_file_unpack2, L.DX_LETTERS_ARROW, img, len(img), width)

    def dexqv(self, quiva: bytes, lossy=False) -> bytes:
        return self._file_call(self.lib.dx_file_dexqv, quiva, len(quiva), int(lossy))

    def qv_use_index(self, d_in, d_seg, n, d_gidx, d_gidx_off, none=0):
        """dx_qv_use_index: a group index made elsewhere (qv_walk(index=True)) for the stream at d_in / d_seg; d_gidx None: take it back"""
        self._chk(self.lib.dx_qv_use_index(self.h, d_in.ptr if d_in else None, d_seg.ptr if d_seg else None, n,
                                           d_gidx.ptr if d_gidx else None, d_gidx_off.ptr if d_gidx_off else None, none))

    def undexqv(self, img: bytes, upper=False) -> bytes:
        return self._file_call(self.lib.dx_file_undexqv, img, len(img), int(upper))

    def qv_subindex(self, on=True):
        self._chk(self.lib.dx_qv_subindex(self.h, int(bool(on))))

    def qv_onepass_info(self) -> dict:
        """dx_qv_onepass_info: the route the last one-pass encode took (groups / direct, scratch, budget)."""
        o = L.OnepassInfo()
        self._chk(self.lib.dx_qv_onepass_info(self.h, C.byref(o)))
        d = {k: int(getattr(o, k)) for k, _ in L.OnepassInfo._fields_ if k not in ("reserved", "chain_waits")}
        d["chain_waits"] = [int(x) for x in o.chain_waits]
        return d

    def trim(self, what=7):
        """dx_trim: give the context's scratch (1), token slots (2), group index (4) back to the device."""
        self._chk(self.lib.dx_trim(self.h, int(what)))

    def set_scratch_budget(self, nbytes: int):
        self._chk(self.lib.dx_set_scratch_budget(self.h, int(nbytes)))

    @staticmethod
    def _sink(sink):
        """ctypes callback around a Python sink.  An exception raised inside a ctypes callback is printed and
        swallowed (the callback then returns 0 = go on), so it is caught here, the library is told to stop
        (it returns DX_E_IO) and the caller re-raises it through `box`."""
        box = []

        def cb(user, data, n, at):
            try:
                return 1 if sink(C.string_at(data, n), at) else 0
            except BaseException as e:                       # noqa: BLE001 -- re-raised by the caller
                box.append(e)
                return 1
        return L.SINK_FN(cb), box

    def unpack2_stream(self, img: bytes, sink, mode=L.DX_LETTERS_LOWER, width=80) -> int:
        """dx_file_unpack2_to: sink(data: bytes, at: int) -> falsy to go on; returns the text's size."""
        total = C.c_size_t()
        cb, box = self._sink(sink)
        rc = self.lib.dx_file_unpack2_to(self.h, int(mode), img, len(img), int(width), cb, None, C.byref(total))
        if box:
            raise box[0]
        self._chk(rc)
        return total.value

    def pack2_stream(self, read, sink, arrow=False, chunk=0) -> int:
        """dx_file_pack2_stream: read(want: int) -> bytes (fewer than want only at the end); sink(data: bytes, at: int) -> falsy
        to go on; returns the image's size."""
        total, line, code = C.c_size_t(), C.c_uint64(), C.c_int()
        cb, box = self._sink(sink)

        def rd(user, buf, want):
            try:
                data = read(want)
                C.memmove(buf, data, len(data))
                return len(data)
            except BaseException as e:                       # noqa: BLE001 -- re-raised by the caller
                box.append(e)
                return -1
        rcb = L.READ_FN(rd)
        rc = self.lib.dx_file_pack2_stream(self.h, int(arrow), rcb, None, int(chunk), cb, None, C.byref(total), C.byref(line), C.byref(code))
        if box:
            raise box[0]
        if rc != 0:
            raise L.DexGPUError(rc, f"line {line.value} (DX_IDX code {code.value}): " + (self.lib.dx_last_error(self.h) or b"").decode())
        return total.value

    def unpack2_pieces(self, read, sink, mode=L.DX_LETTERS_LOWER, width=80, chunk=0) -> int:
        """dx_file_unpack2_stream: read(want: int) -> bytes of the image; sink(data: bytes, at: int); returns the text's size."""
        total = C.c_size_t()
        cb, box = self._sink(sink)

        def rd(user, buf, want):
            try:
                data = read(want)
                C.memmove(buf, data, len(data))
                return len(data)
            except BaseException as e:                       # noqa: BLE001 -- re-raised by the caller
                box.append(e)
                return -1
        rcb = L.READ_FN(rd)
        rc = self.lib.dx_file_unpack2_stream(self.h, int(mode), rcb, None, int(chunk), int(width), cb, None, C.byref(total))
        if box:
            raise box[0]
        self._chk(rc)
        return total.value

    def dexqv_stream(self, quiva: bytes, sink, lossy=False) -> int:
        """dx_file_dexqv_to: sink(data: bytes, at: int) -> falsy to go on; returns the image's size."""
        total, line, code = C.c_size_t(), C.c_uint64(), C.c_int()
        cb, box = self._sink(sink)
        rc = self.lib.dx_file_dexqv_to(self.h, quiva, len(quiva), int(lossy), cb, None, C.byref(total),
                                       C.byref(line), C.byref(code))
        if box:
            raise box[0]
        if rc != 0:
            raise L.DexGPUError(rc, f"line {line.value} (DX_IDX code {code.value}): " + (self.lib.dx_last_error(self.h) or b"").decode())
        return total.value

    def undexqv_stream(self, img: bytes, sink, upper=False) -> int:
        """dx_file_undexqv_plan + dx_file_undexqv_run: sink(data: bytes, at: int) -> falsy to go on; returns the
        text's size.  (The plan alone -- the host walk -- is `undexqv_plan_size`, which needs no GPU.)"""
        plan, total = C.c_void_p(), C.c_size_t()
        keep = C.create_string_buffer(img, len(img))              # must outlive the run
        rc = self.lib.dx_file_undexqv_plan(keep, len(img), C.byref(plan), C.byref(total))
        if rc != 0:
            raise L.DexGPUError(rc, "dx_file_undexqv_plan")
        try:
            cb, box = self._sink(sink)
            rc = self.lib.dx_file_undexqv_run(self.h, plan, int(upper), cb, None)
            if box:
                raise box[0]
            self._chk(rc)
        finally:
            self.lib.dx_file_undexqv_plan_free(plan)
        return total.value


def dexqv_sharded(contexts, quiva: bytes, lossy=False) -> bytes:
    """One .quiva file over several contexts/GPUs (dx_file_dexqv_sharded)."""
    lib = contexts[0].lib
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    out, n, line, code = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_file_dexqv_sharded(arr, len(contexts), quiva, len(quiva), int(lossy), C.byref(out), C.byref(n),
                                   C.byref(line), C.byref(code))
    if rc != 0:
        raise L.DexGPUError(rc, f"line {line.value} (DX_IDX code {code.value})")
    try:
        return C.string_at(out.value, n.value)
    finally:
        lib.dx_file_free(out)


def pack2_sharded(contexts, text: bytes, arrow=False) -> bytes:
    """One .fasta/.arrow file over several contexts/GPUs (dx_file_pack2_sharded)."""
    lib = contexts[0].lib
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    out, n, line, code = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_file_pack2_sharded(arr, len(contexts), int(arrow), text, len(text), C.byref(out), C.byref(n),
                                   C.byref(line), C.byref(code))
    if rc != 0:
        raise L.DexGPUError(rc, f"line {line.value} (DX_IDX code {code.value})")
    try:
        return C.string_at(out.value, n.value)
    finally:
        lib.dx_file_free(out)


# ---- host-only helpers (no GPU needed) ---------------------------------------------------------

def qv_build(hist, tot, params, lossy=False) -> L.QVCoding:
    """Create_QVcoding (QV.c:1029-1169) on the host."""
    lib = L.load()
    h = L.HIST()
    np.ctypeslib.as_array(h)[:] = np.asarray(hist, dtype=np.uint64).reshape(6, 256)
    c = L.QVCoding()
    rc = lib.dx_qv_build(C.byref(h), int(tot), C.byref(params), int(lossy), C.byref(c))
    if rc != 0:
        raise L.DexGPUError(rc, "dx_qv_build")
    return c


def qv_out_bound(hist, n, coding, lossy=False) -> int:
    """dx_qv_out_bound: bytes (without framing) the batch whose raw histograms are `hist` can encode to at most."""
    lib = L.load()
    h = L.HIST()
    np.ctypeslib.as_array(h)[:] = np.asarray(hist, dtype=np.uint64).reshape(6, 256)
    return int(lib.dx_qv_out_bound(C.byref(h), int(n), C.byref(coding), int(lossy)))


def qv_write_coding(coding, prefix: bytes) -> bytes:
    lib = L.load()
    n = C.c_size_t()
    lib.dx_qv_write_coding(C.byref(coding), prefix, len(prefix), None, 0, C.byref(n))
    buf = C.create_string_buffer(n.value)
    rc = lib.dx_qv_write_coding(C.byref(coding), prefix, len(prefix), buf, n.value, C.byref(n))
    if rc != 0:
        raise L.DexGPUError(rc, "dx_qv_write_coding")
    return buf.raw[:n.value]


def qv_read_coding(img: bytes):
    """-> (coding, flip, prefix, consumed)"""
    lib = L.load()
    c, flip, used = L.QVCoding(), C.c_int(), C.c_size_t()
    pre = C.create_string_buffer(len(img) + 1)
    rc = lib.dx_qv_read_coding(img, len(img), C.byref(c), C.byref(flip), pre, len(img) + 1, C.byref(used))
    if rc != 0:
        raise L.DexGPUError(rc, "dx_qv_read_coding")
    return c, flip.value, pre.value, used.value


def qv_walk(img: bytes, index=False):
    """Host boundary walk of a bare .dexqv image -> dict of numpy arrays + coding (QVIndex copy); index=True: with the
    group index of dx_qv_walk_indexed (gidx words, gidx_off, gidx_none)."""
    lib = L.load()
    x = L.QVIndex()
    if isinstance(img, np.ndarray):                       # (a big image: no copy into a bytes object)
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rc = lib.dx_qv_walk_indexed(img.ctypes.data_as(C.c_void_p), img.size, C.byref(x), int(bool(index)))
    else:
        rc = lib.dx_qv_walk_indexed(img, len(img), C.byref(x), int(bool(index)))
    if rc != 0:
        raise L.DexGPUError(rc, "dx_qv_walk")
    try:
        n = x.n
        extra = {}
        if index:
            w = int(x.gidx_words)
            extra = {"gidx": (np.ctypeslib.as_array(x.gidx, (w,)).copy() if w and x.gidx else np.zeros(0, np.uint32)),   # (an image without records: NULL)
                     "gidx_off": np.ctypeslib.as_array(x.gidx_off, (n + 1,)).copy(), "gidx_none": int(x.gidx_none)}
        return {**extra, "n": n,
                "rec_off": np.ctypeslib.as_array(x.rec_off, (n + 1,)).copy(),
                "hdr_off": np.ctypeslib.as_array(x.hdr_off, (n + 1,)).copy(),
                "seg": np.ctypeslib.as_array(x.seg, (max(n, 1), 5))[:n].copy(),
                "len": np.ctypeslib.as_array(x.len, (max(n, 1),))[:n].copy(),
                "hdr4": np.ctypeslib.as_array(x.hdr4, (max(n, 1), 4))[:n].copy(),
                "prefix": bytes(x.prefix), "newv": x.newv, "flip": x.flip,
                "delChar": x.coding.delChar, "subChar": x.coding.subChar}
    finally:
        lib.dx_qv_index_free(C.byref(x))


def frame_headers(hdr4, cnr4=None, lwell=0):
    """-> (blob uint8, off uint64 [n+1], last well)"""
    lib = L.load()
    hdr4 = np.ascontiguousarray(hdr4, dtype=np.int32)
    n = len(hdr4)
    kind = 0 if cnr4 is None else 1
    if cnr4 is not None:
        cnr4 = np.ascontiguousarray(cnr4, dtype=np.uint16)
    bound = lib.dx_frame_bound(hdr4.ctypes.data, n, lwell, kind)
    blob = np.zeros(bound + 16, np.uint8)
    off = np.zeros(n + 1, np.uint64)
    lw = C.c_int32(lwell)
    rc = lib.dx_frame_headers(hdr4.ctypes.data, cnr4.ctypes.data if kind else None, n, kind, C.byref(lw),
                              blob.ctypes.data, off.ctypes.data)
    if rc != 0:
        raise L.DexGPUError(rc, "dx_frame_headers")
    return blob[: int(off[n])], off, lw.value


def index_quiva(text: bytes):
    """-> (off uint64, len uint32, hdr4 int32 [n,4], prefix_len)"""
    lib = L.load()
    cnt, pl, line, code = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_index_quiva(text, len(text), 0, None, None, None, C.byref(cnt), C.byref(pl), C.byref(line), C.byref(code))
    if rc != 0:
        raise L.DexGPUError(rc, f"line {line.value}: DX_IDX code {code.value}")
    n = cnt.value
    off, ln, hdr = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 4), np.int32)
    lib.dx_index_quiva(text, len(text), n, off.ctypes.data, ln.ctypes.data, hdr.ctypes.data, C.byref(cnt),
                       C.byref(pl), C.byref(line), C.byref(code))
    return off, ln, hdr, pl.value


def index_seq(text: bytes, arrow=False):
    """-> (off, tlen, nsym, hdr4, cnr4, prefix_len)"""
    lib = L.load()
    cnt, pl, line, code = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_index_seq(int(arrow), text, len(text), 0, None, None, None, None, None, C.byref(cnt), C.byref(pl),
                          C.byref(line), C.byref(code))
    if rc != 0:
        raise L.DexGPUError(rc, f"line {line.value}: DX_IDX code {code.value}")
    n = cnt.value
    off, tl, ns = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    hdr, cnr = np.zeros((n, 4), np.int32), np.zeros((n, 4), np.uint16)
    lib.dx_index_seq(int(arrow), text, len(text), n, off.ctypes.data, tl.ctypes.data, ns.ctypes.data,
                     hdr.ctypes.data, cnr.ctypes.data, C.byref(cnt), C.byref(pl), C.byref(line), C.byref(code))
    return off, tl, ns, hdr, cnr, pl.value


def undexqv_plan_size(img: bytes) -> int:
    """Size of the text dx_file_undexqv would produce, from the host walk alone (no GPU, no context)."""
    lib = L.load()
    plan, total = C.c_void_p(), C.c_size_t()
    rc = lib.dx_file_undexqv_plan(img, len(img), C.byref(plan), C.byref(total))
    if rc != 0:
        raise L.DexGPUError(rc, "dx_file_undexqv_plan")
    lib.dx_file_undexqv_plan_free(plan)
    return total.value
