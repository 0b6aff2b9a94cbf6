"""Minimal reader of R's RDS serialisation (XDR, version 2/3) -- enough for the reference's example
data files (lists of numeric vectors, numeric matrices).  Test infrastructure."""
import gzip
import struct

import numpy as np


class _R:
    def __init__(self, buf):
        self.b, self.p, self.refs = buf, 0, []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t == 254:                       # NILVALUE
            return None
        if t == 255:                       # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:                         # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 9:                         # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            s = self.b[self.p:self.p + n].decode("latin1")
            self.p += n
            return s
        if t == 2:                         # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                nxt = self.i32()
                nt = nxt & 0xFF
                if nt == 254:
                    break
                if nt != 2:
                    raise ValueError("unexpected pairlist tail")
                has_attr, has_tag = bool(nxt & 0x200), bool(nxt & 0x400)
                _ = attr
            return out
        if t == 238:                       # ALTREP
            info, state, attr = self.item(), self.item(), self.item()
            cls = info[0][1]
            if cls in ("compact_realseq", "compact_intseq"):
                n, start, step = state
                return np.asarray(start + step * np.arange(int(n)), dtype=np.float64)
            if cls == "wrap_real":
                return np.asarray(state[0], dtype=np.float64)
            raise ValueError(f"unsupported ALTREP class {cls}")
        if t in (13, 10):                  # INTSXP / LGLSXP
            n = self.i32()
            v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.p).astype(np.int64)
            self.p += 4 * n
        elif t == 14:                      # REALSXP
            n = self.i32()
            v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.p).astype(np.float64)
            self.p += 8 * n
        elif t == 16:                      # STRSXP
            n = self.i32()
            v = [self.item() for _ in range(n)]
        elif t == 19:                      # VECSXP
            n = self.i32()
            v = [self.item() for _ in range(n)]
        else:
            raise ValueError(f"unsupported SEXP type {t}")
        if has_attr:
            attrs = dict(self.item())
            if "dim" in attrs and isinstance(v, np.ndarray):
                v = v.reshape(tuple(int(x) for x in attrs["dim"]), order="F")
        return v


def read_rds(path):
    with gzip.open(path, "rb") as f:
        buf = f.read()
    assert buf[:2] == b"X\n", "not an XDR RDS file"
    r = _R(buf)
    r.p = 2
    version = r.i32()
    r.i32(); r.i32()
    if version >= 3:
        n = r.i32()
        r.p += n
    return r.item()
