"""Read-only HDF5 reader for ``features.hdf`` (no h5py, no libhdf5): numpy + zlib + one C helper.

The reference stores per-frame feature maps as ``features/<name>`` ``[N, Hf, Wf, C]`` float16, chunked, LZF-compressed,
written with ``h5py.File(..., 'w', libver='latest')`` (/root/reference/scripts/compute_feature_maps.py:160-163, :82-85) and
reads them back with ``hdf[f'features/{features}'][:]`` (/root/reference/autolabel/dataset.py:438-441).  This module follows
the published HDF5 File Format Specification (version 3.0) for exactly what such files — and their ``libver='earliest'``
siblings — contain:

* superblock versions 0-3; object headers version 1 and 2 (with continuation blocks);
* groups: old-style (symbol-table message → B-tree v1 → SNOD → local heap) and new-style compact (link messages);
  dense link storage (fractal heap; > 8 links in a group) raises ``NotImplementedError``;
* dataspace v1/v2 (simple, scalar), datatypes: fixed-point, floating-point, opaque (attributes);
* data layout v3 (compact, contiguous, chunked with a B-tree v1 index) and v4 (chunked: single chunk, implicit,
  fixed array incl. paged data blocks; extensible array / B-tree v2 — resizable datasets — raise);
* filter pipeline v1/v2: deflate (1), shuffle (2), fletcher32 (3), LZF (32000), honouring the per-chunk filter mask;
* attributes v1-v3 stored in the object header (dense attribute storage is skipped).

Pinned: ``tests/golden/hdf5/*.hdf`` were written by h5py 3.3.0 / libhdf5 1.10.6 with the reference's own calls
(``tests/golden/make_hdf5_fixtures.py``); ``tests/test_hdf5_reader.py`` compares every array and attribute.

Interface (the subset of h5py the reference uses): ``File(path, 'r')`` as a context manager, ``f['features/dino']``,
``dataset[:]`` / ``dataset[a:b]`` (leading axis), ``.shape .dtype .chunks .compression .attrs``, ``in``, ``keys()``."""
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF

MSG_DATASPACE, MSG_LINK_INFO, MSG_DATATYPE, MSG_LINK, MSG_LAYOUT, MSG_FILTERS, MSG_ATTRIBUTE = 0x1, 0x2, 0x3, 0x6, 0x8, 0xB, 0xC
MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x10, 0x11
FILTER_DEFLATE, FILTER_SHUFFLE, FILTER_FLETCHER32, FILTER_LZF = 1, 2, 3, 32000


class Hdf5FormatError(ValueError):
    pass


def lzf_decompress_py(src, out_len):
    """liblzf's decoder (the format h5py's filter 32000 wraps), byte for byte.  Slow: used by the tests as the checker for
    the C helper ``aln_lzf_decompress`` and as the decoder when the C library is not built."""
    src = memoryview(src)
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:
            ln = ctrl + 1
            if op + ln > out_len or ip + ln > n:
                raise Hdf5FormatError('lzf: literal run overflows')
            out[op:op + ln] = src[ip:ip + ln]
            ip += ln
            op += ln
        else:
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ln == 7:
                ln += src[ip]
                ip += 1
            ref -= src[ip]
            ip += 1
            ln += 2
            if ref < 0 or op + ln > out_len:
                raise Hdf5FormatError('lzf: bad back reference')
            if ref + ln <= op:
                out[op:op + ln] = out[ref:ref + ln]
            else:                                   # overlapping copy replicates the period
                for k in range(ln):
                    out[op + k] = out[ref + k]
            op += ln
    if op != out_len:
        raise Hdf5FormatError(f'lzf: produced {op} bytes, expected {out_len}')
    return bytes(out)


def lzf_decompress(src, out_len):
    """LZF through the C helper of the C-ABI library when it is built (host code, no GPU needed), else the Python decoder."""
    try:
        from .. import hip
        L = hip.lib()
    except Exception:
        return lzf_decompress_py(src, out_len)
    import ctypes
    src = bytes(src)
    out = ctypes.create_string_buffer(out_len)
    got = L.aln_lzf_decompress(src, len(src), out, out_len)
    if got != out_len:
        raise Hdf5FormatError(f'lzf: produced {got} bytes, expected {out_len}')
    return out.raw


def _unshuffle(buf, size):
    if size <= 1:
        return buf
    a = np.frombuffer(buf, np.uint8)
    n = a.size // size
    body = a[:n * size].reshape(size, n).T.reshape(-1)
    return body.tobytes() + a[n * size:].tobytes()


class _Reader:
    """Little-endian cursor over the file bytes with the superblock's offset / length sizes."""

    def __init__(self, buf, O=8, L=8, base=0):
        self.buf, self.O, self.L, self.base = buf, O, L, base

    def u(self, pos, n):
        return int.from_bytes(self.buf[pos:pos + n], 'little')

    def off(self, pos):
        v = self.u(pos, self.O)
        return UNDEF if v == (1 << (8 * self.O)) - 1 else v + self.base

    def length(self, pos):
        return self.u(pos, self.L)


class _Datatype:
    def __init__(self, r, pos):
        b0 = r.u(pos, 1)
        self.cls, self.version = b0 & 0xF, b0 >> 4
        bits = r.u(pos + 1, 3)
        self.size = r.u(pos + 4, 4)
        order = '>' if bits & 1 else '<'
        if self.cls == 0:                                     # fixed point
            self.dtype = np.dtype(f"{order}{'i' if bits & 8 else 'u'}{self.size}")
        elif self.cls == 1:                                   # floating point
            if self.size not in (2, 4, 8):
                raise NotImplementedError(f'{self.size}-byte float')
            self.dtype = np.dtype(f'{order}f{self.size}')
        elif self.cls == 5:                                   # opaque
            self.dtype = np.dtype(f'V{self.size}')
        elif self.cls == 3:                                   # fixed-length string
            self.dtype = np.dtype(f'S{self.size}')
        else:
            raise NotImplementedError(f'HDF5 datatype class {self.cls}')


def _dataspace(r, pos):
    version, rank, flags = r.u(pos, 1), r.u(pos + 1, 1), r.u(pos + 2, 1)
    if version == 1:
        p = pos + 8
    elif version == 2:
        if r.u(pos + 3, 1) == 2:
            return None                                       # null dataspace
        p = pos + 4
    else:
        raise Hdf5FormatError(f'dataspace version {version}')
    dims = tuple(r.length(p + i * r.L) for i in range(rank))
    maxdims = tuple(r.length(p + (rank + i) * r.L) for i in range(rank)) if flags & 1 else dims
    return dims, maxdims


def _pad8(n):
    return (n + 7) & ~7


class _Object:
    """One object header: the list of (type, flags, data position, size) messages, continuation blocks followed."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        r = f._r
        self.messages = []
        if r.buf[addr:addr + 4] == b'OHDR':
            self._read_v2(addr)
        elif r.u(addr, 1) == 1:
            self._read_v1(addr)
        else:
            raise Hdf5FormatError(f'no object header at {addr}')

    MAX_BLOCKS = 4096      # continuation blocks followed per object header (a corrupt file can chain them in a cycle)

    def _read_v1(self, addr):
        r = self.f._r
        nmsg, size = r.u(addr + 2, 2), r.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and len(self.messages) < nmsg:
            seen += 1
            if seen > self.MAX_BLOCKS:
                raise Hdf5FormatError('object header: too many continuation blocks')
            p, n = blocks.pop(0)
            end = p + n
            end = min(end, len(r.buf))
            while p + 8 <= end and len(self.messages) < nmsg:
                mtype, msize, mflags = r.u(p, 2), r.u(p + 2, 2), r.u(p + 4, 1)
                self._add(mtype, mflags, p + 8, msize, blocks)
                p += 8 + msize

    def _read_v2(self, addr):
        r = self.f._r
        if r.u(addr + 4, 1) != 2:
            raise Hdf5FormatError('object header version')
        flags = r.u(addr + 5, 1)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size = r.u(p, w)
        p += w
        hdr = 4 + (2 if flags & 0x04 else 0)
        blocks = [(p, size)]
        seen = 0
        while blocks:
            seen += 1
            if seen > self.MAX_BLOCKS:
                raise Hdf5FormatError('object header: too many continuation blocks')
            p, n = blocks.pop(0)
            end = min(p + n, len(r.buf))                      # the chunk's checksum follows `end`
            while p + hdr <= end:
                mtype, msize, mflags = r.u(p, 1), r.u(p + 1, 2), r.u(p + 3, 1)
                self._add(mtype, mflags, p + hdr, msize, blocks, v2=True)
                p += hdr + msize

    def _add(self, mtype, mflags, pos, size, blocks, v2=False):
        r = self.f._r
        if mflags & 0x02:
            raise NotImplementedError('shared object header messages')
        if mtype == MSG_CONTINUATION:
            a, n = r.off(pos), r.length(pos + r.O)
            if v2:
                if r.buf[a:a + 4] != b'OCHK':
                    raise Hdf5FormatError('object header continuation signature')
                blocks.append((a + 4, n - 8))
            else:
                blocks.append((a, n))
        self.messages.append((mtype, mflags, pos, size))

    def find(self, mtype):
        return [(p, n) for t, _, p, n in self.messages if t == mtype]

    # -- groups ---------------------------------------------------------------------------------------------------------
    def links(self):
        r = self.f._r
        out = {}
        for p, _ in self.find(MSG_SYMBOL_TABLE):
            self._symbol_table(r.off(p), r.off(p + r.O), out)
        for p, _ in self.find(MSG_LINK_INFO):
            flags = r.u(p + 1, 1)
            q = p + 2 + (8 if flags & 1 else 0)
            if r.off(q) != UNDEF:
                raise NotImplementedError('dense link storage (more than 8 links in one group)')
        for p, _ in self.find(MSG_LINK):
            flags = r.u(p + 1, 1)
            q = p + 2
            ltype = 0
            if flags & 0x08:
                ltype = r.u(q, 1)
                q += 1
            if flags & 0x04:
                q += 8
            if flags & 0x10:
                q += 1
            w = 1 << (flags & 3)
            n = r.u(q, w)
            q += w
            name = bytes(r.buf[q:q + n]).decode('utf-8')
            q += n
            if ltype == 0:
                out[name] = r.off(q)
        return out

    def _symbol_table(self, btree, heap, out):
        r = self.f._r
        if r.buf[heap:heap + 4] != b'HEAP':
            raise Hdf5FormatError('local heap signature')
        data = r.off(heap + 8 + 2 * r.L)

        def name(o):
            return bytes(r.buf[data + o:r.buf.find(b'\0', data + o)]).decode('utf-8')

        def walk(node, depth=0):
            if depth > 32 or len(out) > 1 << 20:
                raise Hdf5FormatError('group B-tree: too deep / too many entries')
            if r.buf[node:node + 4] == b'SNOD':
                n = r.u(node + 6, 2)
                p = node + 8
                for _ in range(n):
                    out[name(r.off(p))] = r.off(p + r.O)
                    p += 2 * r.O + 24
                return
            if r.buf[node:node + 4] != b'TREE' or r.u(node + 4, 1) != 0:
                raise Hdf5FormatError('group B-tree node')
            n = r.u(node + 6, 2)
            p = node + 8 + 2 * r.O + r.L                     # past key 0
            for _ in range(n):
                walk(r.off(p), depth + 1)
                p += r.O + r.L

        walk(btree)

    # -- attributes -----------------------------------------------------------------------------------------------------
    def attributes(self):
        r = self.f._r
        out = {}
        for p, _ in self.find(MSG_ATTRIBUTE):
            version = r.u(p, 1)
            nsz, tsz, ssz = r.u(p + 2, 2), r.u(p + 4, 2), r.u(p + 6, 2)
            q = p + 8 + (1 if version == 3 else 0)
            pad = _pad8 if version == 1 else (lambda n: n)
            name = bytes(r.buf[q:q + nsz]).split(b'\0')[0].decode('utf-8')
            q += pad(nsz)
            try:
                dt = _Datatype(r, q)
            except NotImplementedError:
                continue
            q += pad(tsz)
            space = _dataspace(r, q)
            q += pad(ssz)
            if space is None:
                out[name] = None
                continue
            dims = space[0]
            count = int(np.prod(dims)) if dims else 1
            a = np.frombuffer(bytes(r.buf[q:q + count * dt.size]), dt.dtype)
            out[name] = a.reshape(dims).copy() if dims else a[0]
        return out


class Dataset:
    def __init__(self, f, obj, name):
        self.file, self._obj, self.name = f, obj, name
        r = f._r
        (p, _), = obj.find(MSG_DATATYPE)
        self._dt = _Datatype(r, p)
        self.dtype = self._dt.dtype
        (p, _), = obj.find(MSG_DATASPACE)
        space = _dataspace(r, p)
        self.shape, self.maxshape = space if space is not None else ((), ())
        self._filters = []
        for p, _ in obj.find(MSG_FILTERS):
            self._filters = self._read_filters(p)
        (p, _), = obj.find(MSG_LAYOUT)
        self._read_layout(p)
        self._attrs = None

    # -- metadata -------------------------------------------------------------------------------------------------------
    def _read_filters(self, p):
        r = self.file._r
        version, n = r.u(p, 1), r.u(p + 1, 1)
        q = p + (8 if version == 1 else 2)
        out = []
        for _ in range(n):
            fid = r.u(q, 2)
            q += 2
            nlen = 0
            if version == 1 or fid >= 256:
                nlen = r.u(q, 2)
                q += 2
            flags, nval = r.u(q, 2), r.u(q + 2, 2)
            q += 4
            q += _pad8(nlen) if version == 1 else nlen
            vals = [r.u(q + 4 * i, 4) for i in range(nval)]
            q += 4 * nval
            if version == 1 and nval & 1:
                q += 4
            out.append((fid, flags, vals))
        return out

    def _read_layout(self, p):
        r = self.file._r
        version, cls = r.u(p, 1), r.u(p + 1, 1)
        self.chunks = None
        self._index = None
        if version not in (3, 4):
            raise NotImplementedError(f'data layout version {version}')
        if cls == 0:
            n = r.u(p + 2, 2)
            self._layout = ('compact', p + 4, n)
        elif cls == 1:
            self._layout = ('contiguous', r.off(p + 2), r.length(p + 2 + r.O))
        elif cls == 2 and version == 3:
            nd = r.u(p + 2, 1)
            addr = r.off(p + 3)
            dims = [r.u(p + 3 + r.O + 4 * i, 4) for i in range(nd)]
            self.chunks = tuple(dims[:-1])
            self._layout = ('chunked',)
            self._index = ('btree1', addr)
        elif cls == 2:
            flags, nd, w = r.u(p + 2, 1), r.u(p + 3, 1), r.u(p + 4, 1)
            dims = [r.u(p + 5 + w * i, w) for i in range(nd)]
            q = p + 5 + w * nd
            itype = r.u(q, 1)
            q += 1
            self.chunks = tuple(dims[:-1])
            self._layout = ('chunked',)
            if itype == 1:
                size, mask = None, 0
                if flags & 0x02:
                    size, mask = r.length(q), r.u(q + r.L, 4)
                    q += r.L + 4
                self._index = ('single', r.off(q), size, mask)
            elif itype == 2:
                self._index = ('implicit', r.off(q))
            elif itype == 3:
                self._index = ('farray', r.off(q + 1))
            elif itype == 4:
                raise NotImplementedError('extensible-array chunk index (dataset with an unlimited dimension)')
            elif itype == 5:
                raise NotImplementedError('B-tree v2 chunk index (dataset with several unlimited dimensions)')
            else:
                raise Hdf5FormatError(f'chunk index type {itype}')
        else:
            raise NotImplementedError(f'data layout class {cls}')

    @property
    def compression(self):
        for fid, _, _ in self._filters:
            if fid == FILTER_DEFLATE:
                return 'gzip'
            if fid == FILTER_LZF:
                return 'lzf'
        return None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = self._obj.attributes()
        return self._attrs

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    # -- chunk index ----------------------------------------------------------------------------------------------------
    def _chunk_grid(self):
        return tuple(-(-s // c) for s, c in zip(self.shape, self.chunks))

    def _chunk_bytes(self):
        return int(np.prod(self.chunks)) * self._dt.size

    def _chunk_table(self):
        """{chunk grid coordinate: (address, stored size, filter mask)} for every allocated chunk."""
        r = self.file._r
        kind = self._index[0]
        grid = self._chunk_grid()
        raw = self._chunk_bytes()
        out = {}
        if kind == 'single':
            _, addr, size, mask = self._index
            if addr != UNDEF:
                out[(0,) * len(grid)] = (addr, raw if size is None else size, mask)
        elif kind == 'implicit':
            addr = self._index[1]
            if addr != UNDEF:
                for i, c in enumerate(np.ndindex(*grid)):
                    out[c] = (addr + i * raw, raw, 0)
        elif kind == 'farray':
            self._fixed_array(self._index[1], grid, raw, out)
        elif kind == 'btree1':
            if self._index[1] != UNDEF:
                self._btree1(self._index[1], out)
        return out

    def _fixed_array(self, hdr, grid, raw, out):
        r = self.file._r
        if hdr == UNDEF:
            return
        if r.buf[hdr:hdr + 4] != b'FAHD':
            raise Hdf5FormatError('fixed array header signature')
        client, esize, page_bits = r.u(hdr + 5, 1), r.u(hdr + 6, 1), r.u(hdr + 7, 1)
        nent = r.length(hdr + 8)
        db = r.off(hdr + 8 + r.L)
        if db == UNDEF:
            return
        if r.buf[db:db + 4] != b'FADB':
            raise Hdf5FormatError('fixed array data block signature')
        p = db + 6 + r.O
        per_page = 1 << page_bits
        coords = list(np.ndindex(*grid))
        if nent != len(coords):
            raise Hdf5FormatError(f'fixed array holds {nent} entries for {len(coords)} chunks')

        def element(q, c):
            addr = r.off(q)
            if addr == UNDEF:
                return
            if client == 1:
                w = esize - r.O - 4
                out[c] = (addr, r.u(q + r.O, w), r.u(q + r.O + w, 4))
            else:
                out[c] = (addr, raw, 0)

        if nent <= per_page:
            for i, c in enumerate(coords):
                element(p + i * esize, c)
            return
        npages = -(-nent // per_page)
        bitmap = bytes(r.buf[p:p + (npages + 7) // 8])
        p += (npages + 7) // 8 + 4                           # bitmap, data block checksum; pages follow
        for pg in range(npages):
            n = min(per_page, nent - pg * per_page)
            if bitmap[pg >> 3] & (0x80 >> (pg & 7)):
                for i in range(n):
                    element(p + i * esize, coords[pg * per_page + i])
            p += n * esize + 4                               # each page carries its own checksum

    def _btree1(self, node, out, depth=0):
        r = self.file._r
        if depth > 32:
            raise Hdf5FormatError('chunk B-tree: too deep')
        if r.buf[node:node + 4] != b'TREE' or r.u(node + 4, 1) != 1:
            raise Hdf5FormatError('chunk B-tree node')
        level, n = r.u(node + 5, 1), r.u(node + 6, 2)
        nd = len(self.shape) + 1
        ksize = 8 + 8 * nd
        p = node + 8 + 2 * r.O
        for _ in range(n):
            size, mask = r.u(p, 4), r.u(p + 4, 4)
            offs = [r.u(p + 8 + 8 * i, 8) for i in range(nd - 1)]
            child = r.off(p + ksize)
            if level == 0:
                out[tuple(o // c for o, c in zip(offs, self.chunks))] = (child, size, mask)
            else:
                self._btree1(child, out, depth + 1)
            p += ksize + r.O

    # -- data -----------------------------------------------------------------------------------------------------------
    def _decode_chunk(self, addr, size, mask):
        r = self.file._r
        raw = self._chunk_bytes()
        if addr + size > len(r.buf) or size > 64 * raw + 4096:
            raise Hdf5FormatError(f'chunk at {addr} (+{size} B) lies outside the file')
        buf = bytes(r.buf[addr:addr + size])
        for i in reversed(range(len(self._filters))):
            if mask & (1 << i):
                continue
            fid, _, vals = self._filters[i]
            if fid == FILTER_DEFLATE:
                buf = zlib.decompress(buf)
            elif fid == FILTER_SHUFFLE:
                buf = _unshuffle(buf, vals[0] if vals else self._dt.size)
            elif fid == FILTER_FLETCHER32:
                buf = buf[:-4]
            elif fid == FILTER_LZF:
                buf = lzf_decompress(buf, vals[2] if len(vals) > 2 else raw)
            else:
                raise NotImplementedError(f'HDF5 filter {fid}')
        if len(buf) != raw:
            raise Hdf5FormatError(f'chunk decodes to {len(buf)} bytes, expected {raw}')
        return np.frombuffer(buf, self.dtype).reshape(self.chunks)

    def read(self, start=0, stop=None):
        """Rows ``start:stop`` of the leading axis as a new native-endian array."""
        r = self.file._r
        nbytes = int(np.prod(self.shape, dtype=np.float64)) * self._dt.size if self.shape else self._dt.size
        if nbytes > 1024 * len(r.buf) + (1 << 20):            # (LZF / gzip of real feature maps: 1-3x; constant fill: < 1000x)
            raise Hdf5FormatError(f'dataset shape {self.shape} cannot come from a {len(r.buf)}-byte file')
        if self.chunks is not None and (len(self.chunks) != len(self.shape) or any(c <= 0 for c in self.chunks)):
            raise Hdf5FormatError(f'chunk shape {self.chunks} does not fit dataset shape {self.shape}')
        if not self.shape:
            kind, a, n = self._layout
            return np.frombuffer(bytes(r.buf[a:a + self._dt.size]), self.dtype)[0]
        n0 = self.shape[0]
        stop = n0 if stop is None else min(stop, n0)
        start = max(0, min(start, stop))
        out_shape = (stop - start,) + tuple(self.shape[1:])
        native = self.dtype.newbyteorder('=')
        kind = self._layout[0]
        if kind in ('compact', 'contiguous'):
            _, a, n = self._layout
            if a == UNDEF:
                return np.zeros(out_shape, native)
            row = int(np.prod(self.shape[1:], dtype=np.int64)) * self._dt.size
            a0 = a + start * row
            return np.frombuffer(bytes(r.buf[a0:a0 + (stop - start) * row]), self.dtype).reshape(out_shape).astype(native)
        out = np.zeros(out_shape, native)
        c0 = self.chunks[0]
        for coord, (addr, size, mask) in self._chunk_table().items():
            lo = coord[0] * c0
            if lo >= stop or lo + c0 <= start:
                continue
            chunk = self._decode_chunk(addr, size, mask)
            src, dst = [], []
            for ax, (ci, cs, s) in enumerate(zip(coord, self.chunks, self.shape)):
                a, b = ci * cs, min(ci * cs + cs, s)
                if ax == 0:
                    a2, b2 = max(a, start), min(b, stop)
                    src.append(slice(a2 - a, b2 - a))
                    dst.append(slice(a2 - start, b2 - start))
                else:
                    src.append(slice(0, b - a))
                    dst.append(slice(a, b))
            out[tuple(dst)] = chunk[tuple(src)]
        return out

    def __getitem__(self, key):
        if key is Ellipsis or key == ():
            return self.read()
        if isinstance(key, tuple):
            head, rest = key[0], key[1:]
        else:
            head, rest = key, ()
        if head is Ellipsis:
            return self.read()[key]
        if isinstance(head, slice):
            a, b, step = head.indices(self.shape[0])
            if step == 1:
                out = self.read(a, b)
                return out[(slice(None),) + rest] if rest else out
            return self.read()[key]
        if isinstance(head, (int, np.integer)):
            i = int(head) + (self.shape[0] if head < 0 else 0)
            if not 0 <= i < self.shape[0]:
                raise IndexError(head)
            out = self.read(i, i + 1)[0]
            return out[rest] if rest else out
        return self.read()[key]

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)


class Group:
    def __init__(self, f, obj, name):
        self.file, self._obj, self.name = f, obj, name
        self._links = None

    def _children(self):
        if self._links is None:
            self._links = self._obj.links()
        return self._links

    def keys(self):
        return list(self._children().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._children())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    @property
    def attrs(self):
        return self._obj.attributes()

    def __getitem__(self, path):
        node = self.file if path.startswith('/') else self
        if node is self.file and self is not self.file:
            return self.file[path]
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._children()
            if part not in links:
                raise KeyError(f"Unable to open object (object '{part}' doesn't exist)")
            obj = _Object(self.file, links[part])
            full = (node.name.rstrip('/') + '/' + part)
            if obj.find(MSG_LAYOUT):
                node = Dataset(self.file, obj, full)
            else:
                node = Group(self.file, obj, full)
        return node


class File(Group):
    """``File(path, 'r')`` — read-only; any other mode raises (writing feature files is the extractor's job, out of scope)."""

    def __init__(self, path, mode='r', **_):
        if mode != 'r':
            raise NotImplementedError("autolabel_amd.utils.hdf5 reads only; open with mode 'r'")
        self.filename = str(path)
        self._fh = open(path, 'rb')
        import mmap
        try:
            self._buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise Hdf5FormatError(f'{path}: empty file')
        base = 0
        while self._buf[base:base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self._buf):
                self.close()
                raise Hdf5FormatError(f'{path}: not an HDF5 file (no superblock signature)')
        b = self._buf
        version = b[base + 8]
        if version in (0, 1):
            O, L = b[base + 13], b[base + 14]
            p = base + 24 + (4 if version == 1 else 0)
            r = _Reader(b, O, L)
            r.base = r.u(p, O)
            root = r.off(p + 4 * O + O)                      # root symbol-table entry: name offset, header address
        elif version in (2, 3):
            O, L = b[base + 9], b[base + 10]
            r = _Reader(b, O, L)
            r.base = r.u(base + 12, O)
            root = r.off(base + 12 + 3 * O)
        else:
            self.close()
            raise Hdf5FormatError(f'superblock version {version}')
        self._r = r
        self.superblock_version = version
        Group.__init__(self, self, _Object(self, root), '/')

    def close(self):
        self._links = None
        try:
            self._r = None
            self._buf.close()
        except (BufferError, AttributeError, ValueError):
            pass
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
