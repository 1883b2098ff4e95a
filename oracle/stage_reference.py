"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (pydcop_amd/).

Makes the reference itself (pyDCOP v0.1.2a1, pure Python) available on machines that have no
`/root/reference` -- the GPU box -- WITHOUT putting a single reference source file into this
repository's tree or history:

  stage()   (build container, where /root/reference exists; called by __graft_entry__.build())
            packs `/root/reference/pydcop/**.py` + `/root/reference/tests/instances/*` into ONE
            archive `oracle/_ref/pydcop_reference.tar.gz` + a manifest.  `oracle/_ref/` is
            git-ignored (nothing of it is ever committed) but not gpurun-ignored, so the archive
            travels to the GPU box next to the built `.so` files -- like a compiled reference
            would (`oracle/_ref/` is where the task puts reference build outputs).
  locate()  -> the directory that holds `pydcop/` and `tests/instances/`:
            $PYDCOP_REFERENCE, else /root/reference, else the archive unpacked ONCE into a
            directory private to the current user (~/.cache/pydcop_amd_reference/<digest of the
            archive>, mode 0700, outside the repository; checked for ownership and a completion
            marker before it is reused), else None.

Who may use what this returns: `tests/`, `oracle/ref_harness.py`, `scripts/plugin_on_gpu.sh` (as
the CALLER of the plug-in: the unmodified `pydcop solve` CLI) and `bench.py`'s `cpu_baseline`
leg (the reference's thread-agent runtime timed on the box's host cores, after the timed
region).  Nothing under `pydcop_amd/` looks here.
"""
import hashlib
import io
import json
import os
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(REF_DIR, "pydcop_reference.tar.gz")
MANIFEST = os.path.join(REF_DIR, "pydcop_reference.json")
SOURCE = "/root/reference"


def _members(source):
    out = []
    for top, keep in (("pydcop", lambda n: n.endswith(".py") or n == "pydcop"),
                      (os.path.join("tests", "instances"), lambda n: True)):
        base = os.path.join(source, top)
        for d, dirs, files in os.walk(base):
            dirs[:] = sorted(x for x in dirs if x != "__pycache__")
            for f in sorted(files):
                if keep(f):
                    full = os.path.join(d, f)
                    out.append((os.path.relpath(full, source), full))
    return out


def stage(source=SOURCE, force=False):
    """Pack the reference into oracle/_ref/ (deterministic archive: sorted members, zeroed
    times / owners).  -> manifest dict, or None when `source` is absent (GPU box: the archive
    made in the build container is used as it is)."""
    if not os.path.isdir(os.path.join(source, "pydcop")):
        return None
    members = _members(source)
    h = hashlib.sha256()
    for rel, full in members:
        h.update(rel.encode())
        with open(full, "rb") as f:
            h.update(f.read())
    digest = h.hexdigest()
    if not force and os.path.exists(ARCHIVE) and os.path.exists(MANIFEST):
        try:
            with open(MANIFEST) as f:
                old = json.load(f)
            if old.get("sha256_of_members") == digest:
                return old
        except (OSError, ValueError):
            pass
    os.makedirs(REF_DIR, exist_ok=True)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:gz", compresslevel=6) as tar:
        for rel, full in members:
            info = tar.gettarinfo(full, arcname=rel)
            info.mtime, info.uid, info.gid, info.uname, info.gname = 0, 0, 0, "", ""
            with open(full, "rb") as f:
                tar.addfile(info, f)
    tmp = ARCHIVE + ".tmp"
    with open(tmp, "wb") as f:
        f.write(buf.getvalue())
    os.replace(tmp, ARCHIVE)
    version = ""
    try:
        with open(os.path.join(source, "pydcop", "version.py")) as f:
            import re
            m = re.search(r"__version__\s*=\s*['\"]([^'\"]+)", f.read())
            version = m.group(1) if m else ""
    except OSError:
        pass
    manifest = {"what": "pyDCOP reference (python sources of pydcop/ + tests/instances), packed by "
                        "oracle/stage_reference.py; git-ignored, test infrastructure",
                "source": source, "version": version, "files": len(members),
                "sha256_of_members": digest, "archive_bytes": len(buf.getvalue())}
    with open(MANIFEST, "w") as f:
        json.dump(manifest, f, indent=1)
    return manifest


def _unpacked_root():
    """The archive unpacked into a directory of the CURRENT USER (mode 0700, under
    $XDG_CACHE_HOME or ~/.cache, the system temp directory only as a per-process mkdtemp): a
    directory found there is used only if this user owns it, nobody else can write to it and a
    marker written after a complete unpack names the archive's digest -- nothing another local
    user could have prepared is ever put on sys.path."""
    if not os.path.exists(ARCHIVE):
        return None
    with open(ARCHIVE, "rb") as f:
        tag = hashlib.sha256(f.read()).hexdigest()[:24]
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    cache = os.path.join(base, "pydcop_amd_reference")
    try:
        os.makedirs(cache, mode=0o700, exist_ok=True)
        st = os.stat(cache)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise OSError("cache directory not private")
    except OSError:
        cache = tempfile.mkdtemp(prefix="pydcop_reference_")   # 0700, owned by this process
    root = os.path.join(cache, tag)
    marker = os.path.join(root, ".unpacked")
    if os.path.isdir(os.path.join(root, "pydcop")) and os.path.exists(marker):
        st = os.stat(root)
        with open(marker) as f:
            ok = f.read().strip() == tag
        if ok and st.st_uid == os.getuid() and not (st.st_mode & 0o022):
            return root
        import shutil
        shutil.rmtree(root, ignore_errors=True)
    work = tempfile.mkdtemp(prefix="unpack_", dir=cache)
    with tarfile.open(ARCHIVE, "r:gz") as tar:
        for m in tar.getmembers():   # plain relative files only
            if m.name.startswith(("/", "..")) or ".." in m.name.split("/") or not (m.isfile() or m.isdir()):
                raise RuntimeError(f"unexpected member in {ARCHIVE}: {m.name}")
        tar.extractall(work)
    with open(os.path.join(work, ".unpacked"), "w") as f:
        f.write(tag)
    try:
        os.rename(work, root)
    except OSError:           # another process of this user unpacked it first
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    return root if os.path.isdir(os.path.join(root, "pydcop")) else None


_CACHED = []


def locate():
    """-> directory holding the reference's `pydcop/` package and `tests/instances/`, or None."""
    if _CACHED:
        return _CACHED[0]
    root = None
    env = os.environ.get("PYDCOP_REFERENCE")
    if env and os.path.isdir(os.path.join(env, "pydcop")):
        root = env
    elif os.path.isdir(os.path.join(SOURCE, "pydcop")):
        root = SOURCE
    else:
        root = _unpacked_root()
    _CACHED.append(root)
    return root


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "locate":
        print(locate() or "")
    else:
        m = stage(force="--force" in sys.argv)
        print(json.dumps(m) if m else "no reference checkout at " + SOURCE)
