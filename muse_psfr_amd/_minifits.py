"""Minimal FITS reader/writer (NumPy only) for the SPARTA front end.

The reference does its table I/O with astropy (psfrec.py:1016-1026, 1095-1113).  astropy is not
installed next to PyTorch-ROCm in this image, so this module covers exactly what that code path
needs when astropy is absent: a primary HDU, binary tables with scalar / fixed-size numeric
columns, and image HDUs.  `muse_psfr_amd.psfrec` uses astropy instead whenever it is importable,
and files written here are plain standard FITS (astropy reads them; tests/test_host.py checks
that when an astropy interpreter is available).
"""
import io
import os

import numpy as np

BLOCK = 2880

_TFORM = {'L': 'i1', 'B': 'u1', 'I': '>i2', 'J': '>i4', 'K': '>i8', 'E': '>f4', 'D': '>f8'}
_TFORM_OF = {'b': 'L', 'u1': 'B', 'i2': 'I', 'i4': 'J', 'i8': 'K', 'f4': 'E', 'f8': 'D'}
_BITPIX = {8: 'u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}


class Header:
    """Ordered FITS header: list of (key, value, comment)."""

    def __init__(self, cards=None):
        self.cards = list(cards or [])

    def _find(self, key):
        key = key.upper()
        for i, c in enumerate(self.cards):
            if c[0] == key:
                return i
        return -1

    def __contains__(self, key):
        return self._find(key) >= 0

    def __getitem__(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError(key)
        return self.cards[i][1]

    def get(self, key, default=None):
        i = self._find(key)
        return default if i < 0 else self.cards[i][1]

    def __setitem__(self, key, value):
        comment = ''
        if isinstance(value, tuple):
            value, comment = value
        i = self._find(key)
        if i < 0:
            self.cards.append((key.upper(), value, comment))
        else:
            self.cards[i] = (key.upper(), value, comment)

    def remove(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError(key)
        del self.cards[i]

    def keys(self):
        return [c[0] for c in self.cards]

    def copy(self):
        return Header(self.cards)

    def items(self):
        return [(c[0], c[1]) for c in self.cards]


def _fmt_card(key, value, comment=''):
    if isinstance(value, bool):
        v = '%20s' % ('T' if value else 'F')
    elif isinstance(value, (int, np.integer)):
        v = '%20d' % value
    elif isinstance(value, (float, np.floating)):
        r = repr(float(value)).upper()
        if 'E' not in r and '.' not in r and 'N' not in r:
            r += '.0'
        v = '%20s' % r
    else:
        sv = str(value).replace("'", "''")
        v = "'%-8s'" % sv
    if len(key) > 8:
        card = 'HIERARCH %s = %s' % (key, v.strip())
    else:
        card = '%-8s= %s' % (key, v)
    if comment:
        card += ' / ' + comment
    return ('%-80s' % card)[:80]


def _parse_value(s):
    s = s.strip()
    if not s:
        return None
    if s[0] == "'":
        end = 1
        out = ''
        while end < len(s):
            if s[end] == "'":
                if end + 1 < len(s) and s[end + 1] == "'":
                    out += "'"
                    end += 2
                    continue
                break
            out += s[end]
            end += 1
        return out.rstrip()
    v = s.split('/')[0].strip()
    if v == 'T':
        return True
    if v == 'F':
        return False
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v.replace('D', 'E'))
    except ValueError:
        return v


def _read_header(f):
    cards = []
    while True:
        blk = f.read(BLOCK)
        if len(blk) == 0:
            return None
        if len(blk) < BLOCK:
            raise IOError('truncated FITS header')
        done = False
        for i in range(0, BLOCK, 80):
            c = blk[i:i + 80].decode('ascii', 'replace')
            key = c[:8].strip()
            if key == 'END':
                done = True
                break
            if key == 'HIERARCH' and '=' in c:
                k, v = c[9:].split('=', 1)
                cards.append((k.strip().upper(), _parse_value(v), ''))
            elif c[8:10] == '= ':
                cards.append((key, _parse_value(c[10:]), ''))
        if done:
            return Header(cards)


def _pad(n):
    return (BLOCK - n % BLOCK) % BLOCK


class _HDU:
    def __init__(self, header=None, name=None):
        self.header = header if header is not None else Header()
        if name is not None:
            self.name = name

    @property
    def name(self):
        return str(self.header.get('EXTNAME', ''))

    @name.setter
    def name(self, v):
        self.header['EXTNAME'] = str(v)


class PrimaryHDU(_HDU):
    def __init__(self, header=None):
        super().__init__(header)
        self.data = None

    @property
    def name(self):
        return 'PRIMARY'

    def _serialise(self):
        cards = [_fmt_card('SIMPLE', True, 'conforms to FITS standard'), _fmt_card('BITPIX', 8),
                 _fmt_card('NAXIS', 0), _fmt_card('EXTEND', True)]
        for k, v, c in self.header.cards:
            if k not in ('SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND'):
                cards.append(_fmt_card(k, v, c))
        cards.append('%-80s' % 'END')
        h = ''.join(cards).encode('ascii')
        return h + b' ' * _pad(len(h))

    def copy(self):
        return PrimaryHDU(self.header.copy())


class ImageHDU(_HDU):
    def __init__(self, data=None, header=None, name=None):
        super().__init__(header, name)
        self.data = None if data is None else np.asarray(data)

    def _serialise(self):
        d = self.data
        code = d.dtype.str[1:]
        bitpix = {'u1': 8, 'i2': 16, 'i4': 32, 'i8': 64, 'f4': -32, 'f8': -64}[code]
        cards = [_fmt_card('XTENSION', 'IMAGE', 'Image extension'), _fmt_card('BITPIX', bitpix),
                 _fmt_card('NAXIS', d.ndim)]
        for i, n in enumerate(reversed(d.shape)):
            cards.append(_fmt_card('NAXIS%d' % (i + 1), n))
        cards += [_fmt_card('PCOUNT', 0), _fmt_card('GCOUNT', 1)]
        skip = {'XTENSION', 'BITPIX', 'NAXIS', 'PCOUNT', 'GCOUNT'} | {'NAXIS%d' % i for i in range(1, 10)}
        for k, v, c in self.header.cards:
            if k not in skip:
                cards.append(_fmt_card(k, v, c))
        cards.append('%-80s' % 'END')
        h = ''.join(cards).encode('ascii')
        raw = d.astype(d.dtype.newbyteorder('>')).tobytes()
        return h + b' ' * _pad(len(h)) + raw + b'\0' * _pad(len(raw))

    def copy(self):
        return ImageHDU(None if self.data is None else self.data.copy(), self.header.copy())


class BinTableHDU(_HDU):
    """Binary table; `data` is a NumPy structured array in native byte order."""

    def __init__(self, data=None, header=None, name=None):
        super().__init__(header, name)
        self.data = data

    @classmethod
    def from_columns(cls, columns, header=None, name=None):
        """columns: ordered mapping name -> array of shape (nrows,) or (nrows, k)."""
        names = list(columns)
        arrs = [np.asarray(columns[n]) for n in names]
        nrows = arrs[0].shape[0] if arrs else 0
        dt = []
        for n, a in zip(names, arrs):
            base = a.dtype
            if base.kind == 'b':
                base = np.dtype('bool')
            dt.append((n, base, a.shape[1:]) if a.ndim > 1 else (n, base))
        rec = np.zeros(nrows, dtype=dt)
        for n, a in zip(names, arrs):
            rec[n] = a
        return cls(rec, header, name)

    def _serialise(self):
        d = self.data
        fields, be = [], []
        for n in d.dtype.names:
            ft = d.dtype.fields[n][0]
            base, shape = (ft.subdtype if ft.subdtype else (ft, ()))
            cnt = int(np.prod(shape)) if shape else 1
            code = 'b' if base.kind == 'b' else base.str[1:]
            if code not in _TFORM_OF:
                raise TypeError('unsupported column type %s for %s' % (base, n))
            tf = _TFORM_OF[code]
            fields.append((n, '%d%s' % (cnt, tf), shape))
            be.append((n, _TFORM[tf], shape) if shape else (n, _TFORM[tf]))
        bdt = np.dtype(be)
        raw_arr = np.zeros(d.shape[0], dtype=bdt)
        for n in d.dtype.names:
            col = d[n]
            if col.dtype.kind == 'b':            # FITS logical: 'T' / 'F' bytes
                col = np.where(col, ord('T'), ord('F'))
            raw_arr[n] = col
        raw = raw_arr.tobytes()
        cards = [_fmt_card('XTENSION', 'BINTABLE', 'binary table extension'), _fmt_card('BITPIX', 8),
                 _fmt_card('NAXIS', 2), _fmt_card('NAXIS1', bdt.itemsize),
                 _fmt_card('NAXIS2', d.shape[0]), _fmt_card('PCOUNT', 0), _fmt_card('GCOUNT', 1),
                 _fmt_card('TFIELDS', len(fields))]
        for i, (n, tf, shape) in enumerate(fields, start=1):
            cards.append(_fmt_card('TTYPE%d' % i, n))
            cards.append(_fmt_card('TFORM%d' % i, tf))
            if len(shape) > 1:
                cards.append(_fmt_card('TDIM%d' % i, '(%s)' % ','.join(str(s) for s in reversed(shape))))
        skip = ('XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'PCOUNT', 'GCOUNT', 'TFIELDS')
        for k, v, c in self.header.cards:
            if k in skip or k.startswith(('TTYPE', 'TFORM', 'TDIM')):
                continue
            cards.append(_fmt_card(k, v, c))
        cards.append('%-80s' % 'END')
        h = ''.join(cards).encode('ascii')
        return h + b' ' * _pad(len(h)) + raw + b'\0' * _pad(len(raw))

    def copy(self):
        return BinTableHDU(None if self.data is None else self.data.copy(), self.header.copy())


def _parse_tform(tf):
    tf = tf.strip()
    i = 0
    while i < len(tf) and tf[i].isdigit():
        i += 1
    cnt = int(tf[:i]) if i else 1
    return cnt, tf[i]


def _read_hdu(f, first):
    hdr = _read_header(f)
    if hdr is None:
        return None
    naxis = hdr.get('NAXIS', 0)
    dims = [hdr['NAXIS%d' % i] for i in range(1, naxis + 1)]
    nbytes = abs(hdr.get('BITPIX', 8)) // 8 * int(np.prod(dims)) if dims else 0
    nbytes = (nbytes + hdr.get('PCOUNT', 0)) * hdr.get('GCOUNT', 1) if not first else nbytes
    raw = f.read(nbytes + _pad(nbytes))[:nbytes]
    xt = hdr.get('XTENSION', None)
    if first:
        return PrimaryHDU(hdr)
    if xt == 'BINTABLE':
        be, nat = [], []
        for i in range(1, hdr['TFIELDS'] + 1):
            name = hdr['TTYPE%d' % i]
            cnt, code = _parse_tform(hdr['TFORM%d' % i])
            if code == 'A':
                be.append((name, 'S%d' % cnt))
                nat.append((name, 'S%d' % cnt))
                continue
            if code not in _TFORM:
                raise TypeError('unsupported TFORM %s' % hdr['TFORM%d' % i])
            shape = (cnt,) if cnt > 1 else ()
            td = hdr.get('TDIM%d' % i)
            if td:
                shape = tuple(int(x) for x in reversed(td.strip('() ').split(',')))
            natdt = np.dtype(_TFORM[code]).newbyteorder('=')
            if code == 'L':
                natdt = np.dtype('bool')
            be.append((name, _TFORM[code], shape) if shape else (name, _TFORM[code]))
            nat.append((name, natdt, shape) if shape else (name, natdt))
        arr = np.frombuffer(raw[:hdr['NAXIS1'] * hdr['NAXIS2']], dtype=np.dtype(be))
        out = np.zeros(arr.shape[0], dtype=np.dtype(nat))
        for n in out.dtype.names:
            col = arr[n]
            if out.dtype.fields[n][0].base == np.dtype('bool'):
                col = (col == ord('T'))
            out[n] = col
        return BinTableHDU(out, hdr)
    if xt == 'IMAGE':
        dt = np.dtype(_BITPIX[hdr['BITPIX']])
        data = np.frombuffer(raw, dtype=dt).reshape(tuple(reversed(dims))) if dims else None
        if data is not None:
            data = data.astype(dt.newbyteorder('='))
        return ImageHDU(data, hdr)
    raise TypeError('unsupported extension %r' % xt)


class HDUList(list):
    def __getitem__(self, key):
        if isinstance(key, str):
            for h in self:
                if h.name.upper() == key.upper():
                    return h
            raise KeyError("Extension %r not found." % key)
        return list.__getitem__(self, key)

    def __contains__(self, key):
        if isinstance(key, str):
            return any(h.name.upper() == key.upper() for h in self)
        return list.__contains__(self, key)

    def writeto(self, fileobj, overwrite=False):
        blob = b''.join(h._serialise() for h in self)
        if hasattr(fileobj, 'write'):
            fileobj.write(blob)
            return
        if os.path.exists(fileobj) and not overwrite:
            raise OSError("File %r already exists." % fileobj)
        with io.open(fileobj, 'wb') as f:
            f.write(blob)

    def close(self):
        pass


def open(fileobj):  # noqa: A001 - mirrors astropy.io.fits.open
    close = False
    if hasattr(fileobj, 'read'):
        f = fileobj
    else:
        f = io.open(fileobj, 'rb')
        close = True
    try:
        out = HDUList()
        first = True
        while True:
            h = _read_hdu(f, first)
            if h is None:
                break
            out.append(h)
            first = False
        return out
    finally:
        if close:
            f.close()
