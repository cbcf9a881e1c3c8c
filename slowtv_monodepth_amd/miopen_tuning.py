"""Measured MIOpen solver choices for the convolutions of the BASELINE workloads.

PyTorch asks MIOpen for a solver in "immediate mode"; without measured entries MIOpen either ranks solvers by a heuristic
or (its default hybrid mode) compiles and times candidates during the first steps — ≈45 s of warm-up on a fresh box and,
for this network, picks that are ≈5 % slower than the measured best.  `miopen_db/` holds the user find-db / perf-db that
`scripts/tune_miopen.py` produced on an MI355X (gfx950, 256 CUs) for the cfg-2 networks; pointing MIOPEN_USER_DB_PATH at
a private copy of it makes MIOpen answer from those measurements.  Other shapes are unaffected (no entry -> MIOpen's
default behaviour).  Opt out with SMD_NO_MIOPEN_DB=1 or by exporting MIOPEN_USER_DB_PATH yourself.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import tempfile
from pathlib import Path

__all__ = ['install']

_SRC = Path(__file__).resolve().parent/'miopen_db'


def install() -> str | None:
    """Copy the shipped db files to a per-user scratch dir (MIOpen appends to its user db; the tracked files stay pristine)
    and export MIOPEN_USER_DB_PATH.  Must run before the first convolution of the process.  Returns the directory used."""
    if os.environ.get('SMD_NO_MIOPEN_DB') == '1' or 'MIOPEN_USER_DB_PATH' in os.environ or not _SRC.is_dir(): return None
    files = sorted(_SRC.glob('*.txt'))
    if not files: return None
    tag = hashlib.sha1(b''.join(f.read_bytes() for f in files)).hexdigest()[:10]   # a new shipped db never meets a stale copy
    dst = Path(tempfile.gettempdir())/f'smd_miopen_db_{os.getuid()}_{tag}'
    try:
        dst.mkdir(mode=0o700, parents=True, exist_ok=True)
        st = dst.stat()
        # the directory name is predictable: refuse one that another user owns or can write to (it could be pre-seeded)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022): return None
        for f in files:
            out = dst/f.name
            # MIOpen appends its own measurements to the user db, so an existing copy is kept only if it still STARTS with the
            # shipped content; anything else is replaced
            if out.exists():
                shipped = f.read_bytes()
                with open(out, 'rb') as fh:
                    if fh.read(len(shipped)) == shipped: continue
            tmp = dst/f'.{f.name}.{os.getpid()}'
            shutil.copyfile(f, tmp)
            os.replace(tmp, out)          # atomic: ranks of one node may race here
    except OSError:
        return None
    os.environ['MIOPEN_USER_DB_PATH'] = str(dst)
    return str(dst)
