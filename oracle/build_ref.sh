#!/bin/sh
# Compile the REAL reference kernels (cython/sauvola.pyx, cython/optimiser.pyx)
# from where they lie under /root/reference into oracle/_ref/pyXY/ with the
# reference's own flags (setup.py:7 CFLAGS '-Ofast -DNPY_NO_DEPRECATED_API',
# setup.py:23-26 language_level 3).  Outputs are git-ignored; nothing of the
# reference is copied into the repository (the generated .c is deleted).
# Usage: sh oracle/build_ref.sh [python-interpreter ...]
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
[ -d "$REF/cython" ] || { echo "no reference at $REF: skipping"; exit 0; }
[ $# -gt 0 ] || set -- python3 /opt/conda/bin/python3.9
for PY in "$@"; do
    command -v "$PY" >/dev/null 2>&1 || continue
    "$PY" -c "import Cython, numpy" 2>/dev/null || continue
    TAG=$("$PY" -c "import sys;print('py%d%d'%sys.version_info[:2])")
    OUT="$HERE/_ref/$TAG"
    mkdir -p "$OUT"
    PYINC=$("$PY" -c "import sysconfig;print(sysconfig.get_paths()['include'])")
    NPINC=$("$PY" -W ignore -c "import numpy;print(numpy.get_include())" 2>/dev/null)
    EXT=$("$PY" -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
    for m in sauvola optimiser; do
        "$PY" -W ignore -m cython -3 -o "$OUT/$m.c" "$REF/cython/$m.pyx" >/dev/null 2>&1
        gcc -Ofast -DNPY_NO_DEPRECATED_API -w -shared -fPIC -I"$PYINC" -I"$NPINC" \
            "$OUT/$m.c" -o "$OUT/$m$EXT"
        rm -f "$OUT/$m.c"
    done
    echo "built reference kernels for $TAG in $OUT"
done
