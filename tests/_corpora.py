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
