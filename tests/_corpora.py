"""Seeded corpus generators for the size-tolerance tests (test infrastructure, not product code): the data classes the
round-3 review found outside the stated tolerance -- line-structured logs, JSON records, a SQLite file, a tar stream of
the image's package database -- next to the generators of xz_amd (text, source-tree tar stream) and the oracle (lorem)."""
import json
import os
import random
import sqlite3
import subprocess
import tempfile
import time

_MONTHS = ["Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"]
_B64 = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789+/"


def logs(n, seed=12345):
    """Kernel / daemon log lines: timestamps, 40 hosts, five message shapes with random fields."""
    rnd = random.Random(seed)
    hosts = [f"node{i:03d}" for i in range(40)]
    shapes = [
        lambda: f"kernel: [{rnd.uniform(0, 1e6):12.6f}] usb {rnd.randint(1, 4)}-{rnd.randint(1, 8)}: new high-speed USB device number {rnd.randint(2, 120)} using xhci_hcd",
        lambda: f"sshd[{rnd.randint(300, 65000)}]: Accepted publickey for user{rnd.randint(1, 60)} from 10.{rnd.randint(0, 255)}.{rnd.randint(0, 255)}.{rnd.randint(1, 254)} port {rnd.randint(1024, 65535)} ssh2: RSA SHA256:{''.join(rnd.choice(_B64) for _ in range(43))}",
        lambda: f"systemd[1]: Started Session {rnd.randint(1, 99999)} of user user{rnd.randint(1, 60)}.",
        lambda: f"kernel: [{rnd.uniform(0, 1e6):12.6f}] EXT4-fs (sd{rnd.choice('abcd')}{rnd.randint(1, 4)}): mounted filesystem with ordered data mode. Opts: (null)",
        lambda: f"CRON[{rnd.randint(300, 65000)}]: (root) CMD (   cd / && run-parts --report /etc/cron.{rnd.choice(['hourly', 'daily', 'weekly'])})",
    ]
    out, size, t = [], 0, 1700000000
    while size < n:
        t += rnd.randint(0, 3)
        tm = time.gmtime(t)
        line = f"{_MONTHS[tm.tm_mon - 1]} {tm.tm_mday:2d} {tm.tm_hour:02d}:{tm.tm_min:02d}:{tm.tm_sec:02d} {rnd.choice(hosts)} {rnd.choice(shapes)()}\n"
        out.append(line)
        size += len(line)
    return "".join(out).encode()[:n]


def json_records(n, seed=777):
    """One JSON access-log record per line."""
    rnd = random.Random(seed)
    paths = ["/api/v1/users", "/api/v1/items", "/static/app.js", "/index.html", "/api/v2/search", "/healthz"]
    out, size, t = [], 0, 1700000000
    while size < n:
        r = {"ts": t, "ip": f"10.{rnd.randint(0, 255)}.{rnd.randint(0, 255)}.{rnd.randint(1, 254)}",
             "method": rnd.choice(["GET", "GET", "GET", "POST", "PUT"]),
             "path": rnd.choice(paths) + (f"?id={rnd.randint(1, 100000)}" if rnd.random() < 0.5 else ""),
             "status": rnd.choice([200, 200, 200, 200, 301, 404, 500]), "bytes": rnd.randint(100, 90000),
             "ua": rnd.choice(["Mozilla/5.0 (X11; Linux x86_64)", "curl/7.81.0", "python-requests/2.31"]),
             "rt_ms": round(rnd.expovariate(1 / 30), 2)}
        t += rnd.randint(0, 2)
        line = json.dumps(r) + "\n"
        out.append(line)
        size += len(line)
    return "".join(out).encode()[:n]


def sqlite_file(n, seed=4242):
    """A SQLite database file (names, e-mail addresses, cities, free-text notes, one index), truncated to n bytes."""
    rnd = random.Random(seed)
    first = ["James", "Mary", "John", "Patricia", "Robert", "Jennifer", "Michael", "Linda", "William", "Elizabeth",
             "David", "Barbara", "Richard", "Susan", "Joseph", "Jessica"]
    last = ["Smith", "Johnson", "Williams", "Brown", "Jones", "Garcia", "Miller", "Davis", "Rodriguez", "Martinez",
            "Hernandez", "Lopez"]
    cities = ["Springfield", "Riverside", "Franklin", "Greenville", "Bristol", "Clinton", "Fairview", "Salem", "Madison",
              "Georgetown"]
    words = ("lorem ipsum dolor sit amet consectetur adipiscing elit sed do eiusmod tempor incididunt ut labore et dolore "
             "magna aliqua").split()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "db.sqlite")
        c = sqlite3.connect(path)
        c.execute("create table people(id integer primary key, name text, email text, city text, note text)")
        while True:
            rows = []
            for _ in range(5000):
                f, l = rnd.choice(first), rnd.choice(last)
                rows.append((f"{f} {l}", f"{f.lower()}.{l.lower()}{rnd.randint(1, 999)}@example.com", rnd.choice(cities),
                             " ".join(rnd.choice(words) for _ in range(rnd.randint(5, 40)))))
            c.executemany("insert into people(name,email,city,note) values(?,?,?,?)", rows)
            c.commit()
            if os.path.getsize(path) > n * 0.9:
                break
        c.execute("create index idx_name on people(name)")
        c.commit()
        c.close()
        with open(path, "rb") as fh:
            data = fh.read()
    return data[:n]


def dpkg_tar(n):
    """`tar --sort=name` of the image's package database (real, heterogeneous small files); None when it is not there."""
    root = "/var/lib/dpkg"
    if not os.path.isdir(root):
        return None
    try:
        r = subprocess.run(["tar", "--sort=name", "--mtime=2024-01-01", "--owner=0", "--group=0", "--numeric-owner", "-cf", "-", root],
                           capture_output=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        return None
    data = r.stdout
    if len(data) < (1 << 20):
        return None
    return data[:n]


# ---- literal-heavy / numeric classes the round-4 review found outside the tolerance (numpy, seeded) -------------------
def _np():
    import numpy as np
    return np


def f32_sine(n, seed=101):
    """float32 samples of sin(t * 7e-4) * 50 + N(0, 1e-3): slowly drifting mantissas, almost no matches."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 4 + 1
    t = np.arange(m, dtype=np.float64)
    x = np.sin(t * 7e-4) * 50.0 + rng.normal(0.0, 1e-3, m)
    return x.astype(np.float32).tobytes()[:n]


def f32_two_sines(n, seed=102):
    """float32 sin(t * 1e-3) * 100 + sin(t * 0.0137) * 3 + N(0, 0.01)."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 4 + 1
    t = np.arange(m, dtype=np.float64)
    x = np.sin(t * 1e-3) * 100.0 + np.sin(t * 0.0137) * 3.0 + rng.normal(0.0, 0.01, m)
    return x.astype(np.float32).tobytes()[:n]


def f32_mesh(n, seed=103):
    """float32 xyz vertices of a 512-wide grid, z = sin * cos, noise 1e-4."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 12 + 1
    i = np.arange(m, dtype=np.float64)
    x, y = (i % 512) * 0.125, (i // 512) * 0.125
    z = np.sin(x * 0.31) * np.cos(y * 0.17) * 4.0 + rng.normal(0.0, 1e-4, m)
    v = np.stack([x, y, z], axis=1).astype(np.float32)
    return v.tobytes()[:n]


def fasta_repeats(n, seed=104):
    """FASTA-like ACGT lines of 60 with repeated segments mutated at 1 %."""
    np = _np()
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = np.empty(0, dtype=np.uint8)
    parts, size = [], 0
    pool = []
    while size < n:
        if pool and rng.random() < 0.35:
            src = pool[int(rng.integers(len(pool)))]
            a = int(rng.integers(0, max(1, len(src) - 2000)))
            seg = src[a:a + int(rng.integers(500, 20000))].copy()
            mut = rng.random(len(seg)) < 0.01
            seg[mut] = acgt[rng.integers(0, 4, int(mut.sum()))]
        else:
            seg = acgt[rng.integers(0, 4, int(rng.integers(2000, 40000)))]
            pool.append(seg)
            if len(pool) > 64:
                pool.pop(0)
        parts.append(seg)
        size += len(seg)
    seq = np.concatenate(parts)[: n]
    rows = len(seq) // 60
    body = np.concatenate([seq[: rows * 60].reshape(rows, 60), np.full((rows, 1), 10, dtype=np.uint8)], axis=1).reshape(-1)
    return (b">chr_synthetic\n" + body.tobytes())[:n]


def sparse_text(n, seed=105):
    """Zero pages with islands of lorem-like text."""
    rnd = random.Random(seed)
    words = ("lorem ipsum dolor sit amet consectetur adipiscing elit sed do eiusmod tempor incididunt ut labore et dolore "
             "magna aliqua ut enim ad minim veniam quis nostrud exercitation ullamco laboris nisi aliquip ex ea commodo").split()
    out, size = [], 0
    while size < n:
        z = bytes(rnd.randint(512, 16384))
        txt = " ".join(rnd.choice(words) for _ in range(rnd.randint(20, 600))).encode()
        out.append(z)
        out.append(txt)
        size += len(z) + len(txt)
    return b"".join(out)[:n]


def html_rows(n, seed=106):
    """An HTML table: one <tr> per record with ids, names, prices, dates."""
    rnd = random.Random(seed)
    names = ["widget", "gadget", "sprocket", "flange", "gasket", "bearing", "bracket", "coupling", "valve", "manifold"]
    out, size = ["<html><body><table class=\"inventory\">\n"], 0
    while size < n:
        line = (f"<tr class=\"{'odd' if rnd.random() < 0.5 else 'even'}\"><td>{rnd.randint(1, 10 ** 7)}</td>"
                f"<td><a href=\"/item/{rnd.randint(1, 99999)}\">{rnd.choice(names)}-{rnd.randint(1, 999)}</a></td>"
                f"<td class=\"num\">{rnd.uniform(0.5, 9999):.2f}</td><td>{rnd.randint(2001, 2024)}-{rnd.randint(1, 12):02d}-{rnd.randint(1, 28):02d}</td>"
                f"<td>{rnd.choice(['in stock', 'backorder', 'discontinued'])}</td></tr>\n")
        out.append(line)
        size += len(line)
    return "".join(out).encode()[:n]


def csv_sensors(n, seed=107):
    """CSV rows: timestamp, sensor id, three slowly varying readings, a status word."""
    rnd = random.Random(seed)
    out, size, t = ["ts,sensor,temp_c,rh_pct,press_hpa,status\n"], 0, 1700000000.0
    temp, rh, pr = 21.0, 45.0, 1013.0
    while size < n:
        t += rnd.uniform(0.5, 1.5)
        temp += rnd.gauss(0, 0.02); rh += rnd.gauss(0, 0.05); pr += rnd.gauss(0, 0.03)
        line = f"{t:.3f},S{rnd.randint(1, 32):02d},{temp:.3f},{rh:.2f},{pr:.2f},{rnd.choice(['OK', 'OK', 'OK', 'WARN', 'CAL'])}\n"
        out.append(line)
        size += len(line)
    return "".join(out).encode()[:n]


def pcm16_stereo(n, seed=108):
    """16-bit little-endian stereo PCM: a few drifting partials + noise, the right channel a delayed copy."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 4 + 1
    t = np.arange(m, dtype=np.float64) / 44100.0
    sig = (np.sin(2 * np.pi * 220.0 * t) * 6000 + np.sin(2 * np.pi * 331.7 * t + np.sin(t * 0.9)) * 3500
           + np.sin(2 * np.pi * 1200.5 * t) * 900 * (1 + np.sin(t * 0.31)) + rng.normal(0, 40, m))
    left = sig.astype(np.int16)
    right = (np.roll(sig, 37) * 0.8 + rng.normal(0, 40, m)).astype(np.int16)
    return np.stack([left, right], axis=1).tobytes()[:n]


def f64_sine(n, seed=110):
    """float64 samples of sin(t * 1e-4) * 1000: a period of 62,832 samples (491 KiB).  The class where a far distance -- the same
    phase one period earlier -- is only affordable as a REP and has to survive in the rep stack from piece to piece (round 5:
    +5.4 % vs liblzma before the warm-up walk seeded it)."""
    np = _np()
    return (np.sin(np.arange(n // 8 + 1, dtype=np.float64) * 1e-4) * 1000.0).astype("<f8").tobytes()[:n]


def int32_walk(n, seed=111):
    """int32 random walk, steps in [-1000, 1000)."""
    np = _np()
    rng = np.random.default_rng(seed)
    return np.cumsum(rng.integers(-1000, 1000, n // 4 + 1)).astype("<i4").tobytes()[:n]


def structs24(n, seed=112):
    """24-byte records {u32 id, f32 x (random walk), f32 y (sine), u16 flags, u16 type, i64 timestamp}."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 24 + 1
    rec = np.zeros(m, dtype=[("id", "<u4"), ("x", "<f4"), ("y", "<f4"), ("flags", "<u2"), ("type", "<u2"), ("ts", "<i8")])
    rec["id"] = np.arange(m)
    rec["x"] = np.cumsum(rng.normal(0, 0.01, m)).astype("f4")
    rec["y"] = np.sin(np.arange(m) * 1e-3).astype("f4")
    rec["flags"] = rng.integers(0, 4, m)
    rec["type"] = rng.choice([1, 2, 3, 7], m, p=[0.7, 0.2, 0.05, 0.05])
    rec["ts"] = 1700000000000 + np.cumsum(rng.integers(1, 50, m))
    return rec.tobytes()[:n]


def short_records(k, n, seed=113):
    """k-byte records: 0xA5, a 16-bit counter, k - 3 bytes drawn from {0x00, 0x11, 0x22, 0x33} (2 bits of entropy each) -- the class
    where coding the fields as literals and coding them as far eight-byte matches cost about the same under the respective
    adapted model, and every parse piece picks its own regime (round 5, 24 MiB: k = 13: +4.8 % at preset 6, -3.6 % at 9e; k = 7:
    +2.1 % / +2.6 %; one piece per Block: -4.7 %)."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // k + 1
    r = np.zeros((m, k), dtype=np.uint8)
    r[:, 0] = 0xA5
    r[:, 1] = np.arange(m) & 0xFF
    r[:, 2] = (np.arange(m) >> 8) & 0xFF
    r[:, 3:] = rng.integers(0, 4, (m, k - 3)) * 17
    return r.tobytes()[:n]


def hex_ids(n, seed=114):
    """Lines of random version-4 UUIDs in hex (37 bytes each): four bits per digit, and every match of a good parse is a short rep
    at the line length (the dashes, the '4').  Round 4's scheme: +8.7 % vs liblzma at preset 6 (8 MiB through the oracle); round 5:
    +0.75 %."""
    rnd = random.Random(seed)
    out, size = [], 0
    while size < n:
        line = "%08x-%04x-%04x-%04x-%012x\n" % (rnd.getrandbits(32), rnd.getrandbits(16), 0x4000 | rnd.getrandbits(12),
                                               0x8000 | rnd.getrandbits(14), rnd.getrandbits(48))
        out.append(line)
        size += len(line)
    return "".join(out).encode()[:n]


def reloc_table(n, seed=109):
    """An ELF .rela.dyn-like table: 24-byte records {r_offset, r_info, r_addend} (little-endian u64 each): offsets that grow by 8
    with occasional jumps, R_X86_64_RELATIVE almost always (a few GLOB_DAT / 64 with a symbol index), addends that wander
    through a text segment.  Fixed-size records = "distance 24" thousands of times in a row: the class where a model that
    has learnt "rep0" and one that has learnt "match, distance 24" are both stable (round 5: +8.5 % vs liblzma on the real
    section of libMIOpen.so before the coder kept the parser's rep / match choice)."""
    np = _np()
    rng = np.random.default_rng(seed)
    m = n // 24 + 1
    step = np.where(rng.random(m) < 0.97, 8, rng.integers(16, 4096, m) & ~7)
    off = 0x3E4E3A50 + np.cumsum(step)
    kind = rng.random(m)
    info = np.where(kind < 0.96, 8, np.where(kind < 0.98, 6 | (rng.integers(1, 20000, m) << 32), 1 | (rng.integers(1, 20000, m) << 32)))
    add = 0x31E29540 + np.cumsum(rng.integers(-200, 1400, m)) * 16
    add = np.where(info == 8, add, 0)
    rec = np.stack([off, info, add], axis=1).astype("<u8")
    return rec.tobytes()[:n]



# ---- classes the round-5 review probed and found outside the tolerance (image pixels, varint records, a cycled source tree,
# CJK text): generators restated from the review's descriptions, seeded, no dependence on the image except `cycled_tree` ------
def rgba_image(n, seed=120):
    """1024-pixel-wide RGBA8 image: R = (x // 8 * 8) % 256, G = (y // 8 * 8) % 256, B = ((x // 32) ^ (y // 32)) * 16 % 256, each
    + uniform {0, 1, 2} noise, A = 255.  Round 5: +10.9 % vs liblzma at preset 6 with ~190 parse pieces per Block, +0.7 % with 3."""
    np = _np()
    rng = np.random.default_rng(seed)
    rows = n // 4096 + 1
    y, x = np.meshgrid(np.arange(rows), np.arange(1024), indexing="ij")
    px = np.empty((rows, 1024, 4), dtype=np.uint8)
    px[..., 0] = ((x // 8 * 8) % 256 + rng.integers(0, 3, x.shape)) & 255
    px[..., 1] = ((y // 8 * 8) % 256 + rng.integers(0, 3, x.shape)) & 255
    px[..., 2] = (((x // 32) ^ (y // 32)) * 16 % 256 + rng.integers(0, 3, x.shape)) & 255
    px[..., 3] = 255
    return px.tobytes()[:n]


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def varint_records(n, seed=121):
    """Length-prefixed protobuf-style records: field 1 varint millisecond timestamp (+U(0, 2000) per record), field 2 varint
    0..300, field 3 one of five short strings, field 4 a float32 gaussian.  Round 5: +3.8 % vs liblzma at preset 6."""
    import struct
    rnd = random.Random(seed)
    names = [b"temperature", b"humidity", b"pressure_hpa", b"wind", b"battery_mv"]
    out, size, t = [], 0, 1700000000000
    while size < n:
        t += rnd.randint(0, 2000)
        s = rnd.choice(names)
        body = (b"\x08" + _varint(t) + b"\x10" + _varint(rnd.randint(0, 300)) + b"\x1a" + _varint(len(s)) + s
                + b"\x25" + struct.pack("<f", rnd.gauss(0.0, 1.0)))
        rec = _varint(len(body)) + body
        out.append(rec)
        size += len(rec)
    return b"".join(out)[:n]


def cjk_text(n, seed=122):
    """Zipf-distributed text of 8,000 words of one to three CJK characters (U+4E00 ...), UTF-8, a space between words and a
    full stop + newline now and then.  Round 5: +1.9 % vs liblzma at preset 6."""
    np = _np()
    rng = np.random.default_rng(seed)
    rnd = random.Random(seed)
    vocab = ["".join(chr(0x4E00 + rnd.randrange(0x5000)) for _ in range(rnd.randint(1, 3))).encode("utf-8") for _ in range(8000)]
    out, size = [], 0
    while size < n:
        idx = np.minimum(rng.zipf(1.3, 4096) - 1, 7999)
        for k, i in enumerate(idx):
            w = vocab[int(i)]
            out.append(w)
            out.append(b"\xe3\x80\x82\n" if k % 17 == 16 else b" ")
            size += len(w) + 1
    return b"".join(out)[:n]


def cycled_tree(n, seed=123):
    """A source tree cycled to the Block length (the review's `xzsrc`: /root/reference's src/**/*.[ch] + *.txt + po/*.po, ratio
    0.019, +2.2 % vs liblzma at preset 6).  /root/reference does not exist on the GPU box, so the tree is this repository's own
    sources and documents plus the interpreter's top-level library modules -- about 3 MiB per cycle as well."""
    import glob
    import sysconfig
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for pat in ("xz_amd/csrc/*.c", "xz_amd/csrc/*.h", "xz_amd/csrc/*.hip", "oracle/*.c", "oracle/*.h", "tests/*.py", "*.md", "include/*.h"):
        files += sorted(glob.glob(os.path.join(root, pat)))
    files += sorted(glob.glob(os.path.join(sysconfig.get_paths()["stdlib"], "*.py")))[:120]
    parts, size = [], 0
    for f in files:
        try:
            with open(f, "rb") as fh:
                b = fh.read()
        except OSError:
            continue
        parts.append(b)
        size += len(b)
    tree = b"".join(parts)
    if len(tree) < (1 << 20):
        return None
    return (tree * (n // len(tree) + 1))[:n]


REVIEW_CLASSES = {"rgba": rgba_image, "varint": varint_records, "cjk": cjk_text, "cycled_tree": cycled_tree}


# ---- a seeded random CLASS generator (round-5 review: "so that the tolerance is a property, not a list"): tables of records of
# width 1 .. 64 whose fields are drawn from {counter, random walk, enum, noise bits, text, constant}, optionally a mixture of two
# such tables in alternating segments.  Integer arithmetic only (numpy Generator.integers: the streams do not depend on libm).
_WORDS = (b"the of and to in is that for it as was with be by on not he this are or his from at which but have an had they you were "
          b"their one all we can her has there been if more when will would who so no said what up its about than into them").split()


def _field(rng, kind, nrec, fw):
    """(nrec, fw) uint8: one field of the table"""
    np = _np()
    if kind == 0:                                     # counter: start + step * i, little or big endian
        v = int(rng.integers(0, 1 << 30)) + int(rng.integers(1, 1 << int(rng.integers(1, 12)))) * np.arange(nrec, dtype=np.uint64)
    elif kind == 1:                                   # random walk with small steps
        k = 1 << int(rng.integers(1, 10))
        v = (int(rng.integers(0, 1 << 30)) + np.cumsum(rng.integers(-k, k + 1, nrec))).astype(np.uint64)
    elif kind == 2:                                   # enum: one of m values, skewed
        m = int(rng.integers(2, 17))
        vals = rng.integers(0, 1 << 62, m).astype(np.uint64)
        idx = np.minimum(rng.integers(0, m, nrec), rng.integers(0, m, nrec))
        v = vals[idx]
    elif kind == 3:                                   # noise in the low b bits of every byte, the rest constant
        b = int(rng.integers(1, 9))
        base = rng.integers(0, 256, fw).astype(np.uint8) & np.uint8((0xFF << b) & 0xFF)
        return base[None, :] | rng.integers(0, 1 << b, (nrec, fw)).astype(np.uint8)
    elif kind == 4:                                   # text: words from a small vocabulary, cut / padded to the field
        joined = b" ".join(_WORDS[int(i)] for i in rng.integers(0, len(_WORDS), 4096))
        buf = np.frombuffer((joined * ((nrec * fw) // len(joined) + 2))[:nrec * fw + 4096], dtype=np.uint8)
        off = int(rng.integers(0, 4096))
        return buf[off:off + nrec * fw].reshape(nrec, fw).copy()
    else:                                             # constant
        return np.broadcast_to(rng.integers(0, 256, fw).astype(np.uint8), (nrec, fw)).copy()
    by = (v[:, None] >> (8 * np.arange(fw, dtype=np.uint64))[None, :]).astype(np.uint8)
    return by[:, ::-1] if int(rng.integers(0, 2)) else by


def _table(rng, n):
    np = _np()
    width = int(rng.integers(1, 65))
    nrec = n // width + 1
    cols, col = [], 0
    while col < width:
        fw = min(width - col, int(rng.choice([1, 1, 2, 2, 4, 4, 8, 3, 6, 12])))
        cols.append(_field(rng, int(rng.integers(0, 6)), nrec, fw))
        col += fw
    return np.concatenate(cols, axis=1).tobytes()[:n]


def random_class(seed, n):
    """Draw number `seed` of the random class generator: n bytes."""
    np = _np()
    rng = np.random.default_rng(1_000_003 * (seed + 1))
    a = _table(rng, n)
    if int(rng.integers(0, 3)) != 0:
        return a
    b = _table(rng, n)                                # a mixture: alternating segments of two tables
    seg = int(rng.integers(64, 2049)) << 10
    out, pos, which = [], 0, 0
    while pos < n:
        out.append((b if which else a)[pos:pos + seg])
        pos += seg
        which ^= 1
    return b"".join(out)[:n]

NUMERIC_CLASSES = {
    "f32sine": f32_sine, "f32two": f32_two_sines, "f32mesh": f32_mesh, "fasta": fasta_repeats, "sparse": sparse_text,
    "html": html_rows, "csv": csv_sensors, "pcm16": pcm16_stereo, "f64sine": f64_sine, "int32walk": int32_walk, "structs24": structs24, "hexids": hex_ids,
    "rec7": lambda n: short_records(7, n),          # (known outside in round 5: +1.66 / +2.85 %; round 6: +0.04 / +0.60 %)
}
# Classes known to lie OUTSIDE the stated tolerance, kept in the tests so that the number is measured and pinned, not hidden:
# relocs (preset 6, 24 MiB: +4.9 %): liblzma settles into coding every record as an 11-byte match (the constant r_info + two
# addend bytes) with the NEAREST earlier record that shares them -- BT4 returns the nearest match of every length; the
# suffix-neighbourhood finder has "nearest" heads at 8 and 16 bytes only, its 11-byte candidates are the most recent of ten
# suffix neighbours (about 3 bits farther), so the parser stays with rep0 + two literals.  The real .rela.dyn section of
# libMIOpen.so (24-byte records with more regular addends) is inside: +2.1 %.
# rec13 / rec7 (short_records): regime-sensitive, see its docstring.
# measured and pinned with their own bound (round 6: +3.90 / +3.28 % and +3.61 / -4.67 % at presets 6 / 9e; round 5: +4.57 and +5.77 %):
# the parser's path through the model's equilibria, not the pieces -- DESIGN.md section 5, "tables of records"
KNOWN_OUTSIDE = {"relocs": (reloc_table, 0.045), "rec13": (lambda n: short_records(13, n), 0.045)}


def elf_metadata(n):
    """The first n bytes of the largest ELF shared object of the ROCm install (libMIOpen.so: .dynsym, .dynstr, 12 MiB of
    .rela.dyn, .gcc_except_table, the start of .rodata): tables of fixed-size records.  Image-dependent: None when absent."""
    import glob
    files = [f for f in glob.glob("/opt/rocm/lib/libMIOpen.so.*") if os.path.isfile(f) and not os.path.islink(f)]
    if not files or os.path.getsize(sorted(files)[0]) < n:
        return None
    with open(sorted(files)[0], "rb") as fh:
        return fh.read(n)
