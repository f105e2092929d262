"""On-disk model container for `vw-model_final.model` (the name opal.py:298,305 gives the regressor and
opal.py:102-109 looks up by the `*_final.model` glob).

The reference's file is whatever `vw -f` writes; VW is absent here (parity unpinned), so the primary format is a
native container that holds exactly what the scoring kernel needs: bits, K (--oaa), the VW switches, and the
compact weight table either dense (2^bits float32) or as (slot, weight) pairs when that is smaller.

layout (little endian):  magic "OPALHIP1" | u32 version | u32 bits | u32 K | u32 flags | f32 min_label | f32 max_label
                         | u64 nnz | u32 dense(0/1) | u32 batches | payload [| resume section]
                         payload dense : float32[2^bits]          payload sparse: uint32 slot[nnz], float32 weight[nnz]
flags: bit0 hash_all, bit1 add_constant, bit2 clip
version 1: weights only (what predict needs; the default, files as small as they can be).
version 2 (written by `train --save-resume`, the counterpart of vw's --save_resume at reference opal.py:310): the payload is
followed by the learner's state, so that training can go on where it stopped:
                         magic "OPALRES2" | u64 g_nnz | uint32 slot[g_nnz] | float32 G[g_nnz]      (AdaGrad accumulators, sparse)
and `batches` counts the training batches the model has seen (a resumed run draws its next batch with seed 421 + batches).
"""
import struct

import numpy as np

MAGIC = b"OPALHIP1"
_HDR = struct.Struct("<8sIIIIffQII")

__all__ = ["save_model", "load_model", "read_header", "import_vw_readable_model", "export_vw_readable_model",
           "read_vw_binary_header"]


def _count_nonzero(w, chunk=1 << 27):
    """Number of non-zero entries, scanned in chunks (on the device for a torch tensor)."""
    n = w.shape[0]
    if isinstance(w, np.ndarray):
        return int(sum(int(np.count_nonzero(w[s:s + chunk])) for s in range(0, n, chunk)))
    return int(sum(int((w[s:s + chunk] != 0).sum().item()) for s in range(0, n, chunk)))


def _nonzero_chunks(w, chunk=1 << 27):
    """(slot uint32, weight float32) of the non-zero entries of a 1-D tensor/array, scanning in chunks (torch.nonzero
    refuses tensors with more than INT_MAX elements).  For a device tensor only the compacted pairs cross PCIe."""
    idx_parts, val_parts = [], []
    n = w.shape[0]
    is_np = isinstance(w, np.ndarray)
    for s in range(0, n, chunk):
        part = w[s:s + chunk]
        if is_np:
            nz = np.flatnonzero(part)
            idx_parts.append((nz + s).astype(np.uint32))
            val_parts.append(part[nz].astype(np.float32))
        else:
            nz = part.nonzero().squeeze(1)
            vals = part[nz]
            # slots < 2^32: ship them as 4-byte integers (bit pattern of uint32 in an int32 tensor for s >= 2^31)
            idx_parts.append((nz + s).cpu().numpy().astype(np.uint32))
            val_parts.append(vals.cpu().numpy())
    return (np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.uint32),
            np.concatenate(val_parts) if val_parts else np.zeros(0, np.float32))


RESUME_MAGIC = b"OPALRES2"


def save_model(path, w, K, bits, hash_all=False, add_constant=True, clip=True, min_label=-1.0, max_label=1.0, g=None, batches=0):
    """w: the 2^bits table (numpy float32 or torch tensor, any device).  g (optional): the learner's 2^bits accumulators;
    with it the file is a version-2 container that training can resume from.  Returns the number of non-zero weights."""
    assert w.shape[0] == (1 << bits)
    assert g is None or g.shape[0] == (1 << bits)
    nnz = _count_nonzero(w)
    dense = nnz * 8 >= (1 << bits) * 4
    flags = int(bool(hash_all)) | (int(bool(add_constant)) << 1) | (int(bool(clip)) << 2)
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, 1 if g is None else 2, bits, K, flags, min_label, max_label, nnz, int(dense), int(batches) if g is not None else 0))
        if dense:
            chunk = 1 << 27                                    # 512 MiB pieces: bounded host memory for 8 GiB tables
            for s in range(0, w.shape[0], chunk):
                part = w[s:s + chunk]
                arr = part if isinstance(part, np.ndarray) else part.cpu().numpy()
                np.ascontiguousarray(arr, dtype=np.float32).tofile(f)
        else:
            idx, val = _nonzero_chunks(w)
            idx.tofile(f)
            val.tofile(f)
        if g is not None:
            gidx, gval = _nonzero_chunks(g)
            f.write(RESUME_MAGIC + struct.pack("<Q", gidx.size))
            gidx.tofile(f)
            gval.tofile(f)
    return nnz


def read_header(path):
    with open(path, "rb") as f:
        raw = f.read(_HDR.size)
    if len(raw) < _HDR.size or raw[:8] != MAGIC:
        raise RuntimeError("%s is not an opal_amd model (a VW binary regressor cannot be read: no vw to pin its layout)" % path)
    magic, version, bits, K, flags, minl, maxl, nnz, dense, batches = _HDR.unpack(raw)
    if version not in (1, 2):
        raise RuntimeError("unsupported opal_amd model version %d" % version)
    return dict(bits=bits, K=K, hash_all=bool(flags & 1), add_constant=bool(flags & 2), clip=bool(flags & 4),
                min_label=minl, max_label=maxl, nnz=nnz, dense=bool(dense), version=version, resumable=version == 2,
                batches=batches if version == 2 else 0)


def model_kind(path):
    """'opal' (native container), 'vw-text' (a `--readable_model` dump) or 'vw-binary' (anything else: what `vw -f` writes)."""
    with open(path, "rb") as f:
        head = f.read(16)
    if head[:8] == MAGIC:
        return "opal"
    if head[:8].lower() == b"version ":
        return "vw-text"
    return "vw-binary"


def load_model(path, engine, resume=False):
    """Loads the model into the engine's GPU context (weights + VW switches) and returns the header dict.  Dispatches on
    the file's first bytes: the native container, a VW `--readable_model` text dump (imported, VW-unverified), or a VW
    binary regressor, whose header is parsed as far as it can be and then refused with what was found.
    resume=True also installs the learner's accumulators (version-2 container only: RuntimeError otherwise), so that
    engine.train_begin(K, bits, resume=True) continues the training the file was saved from."""
    kind = model_kind(path)
    if resume and kind != "opal":
        raise RuntimeError("%s holds no resume state (only a model written by `train --save-resume` does)" % path)
    if kind == "vw-text":
        m = import_vw_readable_model(path)
        engine.load_weights_sparse(m["idx"], m["val"], m["K"], m["bits"])
        engine.set_vw_options(False, True, True, m["min_label"], m["max_label"])
        return dict(bits=m["bits"], K=m["K"], hash_all=False, add_constant=True, clip=True, min_label=m["min_label"],
                    max_label=m["max_label"], nnz=int(m["idx"].size), dense=False)
    if kind == "vw-binary":
        read_vw_binary_header(path)        # always raises, with the fields it could parse
    h = read_header(path)
    with open(path, "rb") as f:
        f.seek(_HDR.size)
        if h["dense"]:
            w = np.fromfile(f, dtype=np.float32, count=1 << h["bits"])
            if w.size != (1 << h["bits"]):
                raise RuntimeError("truncated model file " + path)
            engine.load_weights(w, h["K"], h["bits"])
        else:
            idx = np.fromfile(f, dtype=np.uint32, count=h["nnz"])
            val = np.fromfile(f, dtype=np.float32, count=h["nnz"])
            if idx.size != h["nnz"] or val.size != h["nnz"]:
                raise RuntimeError("truncated model file " + path)
            engine.load_weights_sparse(idx, val, h["K"], h["bits"])
        if resume:
            if not h["resumable"]:
                raise RuntimeError("%s holds no resume state (only a model written by `train --save-resume` does)" % path)
            sec = f.read(16)
            if len(sec) != 16 or sec[:8] != RESUME_MAGIC:
                raise RuntimeError("truncated model file " + path)
            g_nnz = struct.unpack("<Q", sec[8:])[0]
            gidx = np.fromfile(f, dtype=np.uint32, count=g_nnz)
            gval = np.fromfile(f, dtype=np.float32, count=g_nnz)
            if gidx.size != g_nnz or gval.size != g_nnz:
                raise RuntimeError("truncated model file " + path)
            engine.load_accumulators_sparse(gidx, gval)
    engine.set_vw_options(h["hash_all"], h["add_constant"], h["clip"], h["min_label"], h["max_label"])
    return h


def import_vw_readable_model(path):
    """Best-effort reader of a VW `--readable_model` text dump.  UNVERIFIED against a real vw
    (none exists here: parity unpinned); follows the published VW 8.x layout: header lines `key:value` (`bits:`,
    `Min label:`, `Max label:`, `options: ... --oaa K ...`), a line `:<n>` that ends the header, then one
    `slot:weight[ adaptive normalized]` line per non-zero weight, slot = (h*K + c) mod 2^bits.
    Returns dict(idx uint32[], val float32[], K, bits, min_label, max_label)."""
    import re
    bits = K = None
    minl, maxl = -1.0, 1.0
    idx, val = [], []
    in_weights = False
    with open(path, "r", errors="replace") as f:
        for line in f:
            line = line.strip()
            if not in_weights:
                low = line.lower()
                if low.startswith("bits:"):
                    bits = int(line.split(":", 1)[1])
                elif low.startswith("min label:"):
                    minl = float(line.split(":", 1)[1])
                elif low.startswith("max label:"):
                    maxl = float(line.split(":", 1)[1])
                elif low.startswith("options:"):
                    m = re.search(r"--oaa[ =](\d+)", line)
                    if m:
                        K = int(m.group(1))
                elif re.fullmatch(r":\d+", line):
                    in_weights = True
                continue
            m = re.match(r"(\d+):(\S+)", line)
            if m:
                idx.append(int(m.group(1)))
                val.append(float(m.group(2)))
    if bits is None or K is None:
        raise RuntimeError("%s: not a VW readable model with `bits:` and `--oaa K`" % path)
    return dict(idx=np.array(idx, dtype=np.uint32), val=np.array(val, dtype=np.float32), K=K, bits=bits,
                min_label=minl, max_label=maxl)


def export_vw_readable_model(path, w, K, bits, min_label=-1.0, max_label=1.0, version="8.6.1"):
    """Writes the compact table as a VW `--readable_model` text dump in the layout import_vw_readable_model reads (so the
    two round-trip), for the day a real vw is at hand: the harness (tests/test_vw_parity.py) can then diff one of OUR models,
    slot by slot, against the `--readable_model` dump of a vw trained on the same stream (vw itself loads only binary
    regressors, so this is the direction text can travel).  VW-unverified like the importer.  Returns the number of lines."""
    idx, val = _nonzero_chunks(w)
    with open(path, "w") as f:
        f.write("Version %s\nId \nMin label:%s\nMax label:%s\nbits:%d\nlda:0\n0 ngram:\n0 skip:\n"
                "options: --oaa %d\nChecksum: 0\n:0\n" % (version, repr(float(min_label)), repr(float(max_label)), bits, K))
        for a in range(0, idx.size, 1 << 20):
            f.write("".join("%d:%s\n" % (i, repr(float(v))) for i, v in zip(idx[a:a + (1 << 20)].tolist(), val[a:a + (1 << 20)].tolist())))
    return int(idx.size)


def read_vw_binary_header(path):
    """[VW-unverified] Parses the start of a VW 8.x binary regressor (what `vw -f model` at reference opal.py:309 writes)
    as far as its layout is known from VW's published save_load_header: u32 length + version string; u32 length + id
    string; the byte 'm'; f32 min_label; f32 max_label; u32 bits; u32 lda; u32 ngram count (+ that many 3-byte strings);
    u32 skip count (+ strings); u32 length + options string; u32 checksum.  It stops at the first field that is not
    plausible and ALWAYS raises RuntimeError with the fields read so far: the weight section of a `--save_resume`
    regressor (per-slot adaptive/normalised state, stride) cannot be verified without a vw binary, so it is never loaded."""
    fields = {}
    with open(path, "rb") as f:
        data = f.read(1 << 16)
    pos = 0

    def stop(why):
        got = ", ".join("%s=%r" % kv for kv in fields.items()) or "nothing"
        raise RuntimeError("%s is not an opal_amd model; as a VW binary regressor its header reads: %s; stopped at offset %d: %s. "
                           "A VW regressor cannot be loaded (no vw here to pin its weight layout): re-train with opal_amd, or "
                           "dump it with `vw --readable_model` and load that text file." % (path, got, pos, why))

    def u32():
        nonlocal pos
        if pos + 4 > len(data):
            stop("file ends inside a 32-bit field")
        v = struct.unpack_from("<I", data, pos)[0]
        pos += 4
        return v

    def string(what, limit):
        nonlocal pos
        n = u32()
        if n > limit or pos + n > len(data):
            stop("%s length %d is not plausible" % (what, n))
        raw = data[pos:pos + n]
        pos += n
        return raw.rstrip(b"\0").decode("latin-1")

    fields["version"] = string("version string", 64)
    if not fields["version"] or not all(ch.isdigit() or ch == "." for ch in fields["version"]):
        stop("version string is not of the form 8.x.y")
    fields["id"] = string("id string", 4096)
    if pos >= len(data) or data[pos:pos + 1] != b"m":
        stop("expected the model marker byte 'm'")
    pos += 1
    if pos + 8 > len(data):
        stop("file ends inside min/max label")
    fields["min_label"], fields["max_label"] = struct.unpack_from("<ff", data, pos)
    pos += 8
    if not (np.isfinite(fields["min_label"]) and np.isfinite(fields["max_label"])):
        stop("min/max label are not finite")
    fields["bits"] = u32()
    if not 1 <= fields["bits"] <= 32:
        stop("bits is outside 1..32")
    fields["lda"] = u32()
    for what in ("ngram", "skip"):
        cnt = u32()
        if cnt > 256:
            stop("%s count %d is not plausible" % (what, cnt))
        fields[what] = [data[pos + 3 * i:pos + 3 * i + 3].rstrip(b"\0").decode("latin-1") for i in range(cnt)]
        pos += 3 * cnt
    fields["options"] = string("options string", 1 << 15)
    fields["checksum"] = u32()
    stop("header complete; the weight section is not read")
