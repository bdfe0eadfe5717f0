#!/bin/bash
# Builds the one compiled piece of the reference that the SIV score depends on - the Cython TriangleHash of
# dev_fn/external/libmesh - from the reference's OWN source, where it lies under /root/reference, into oracle/_ref/
# (git-ignored; test infrastructure only: it lets oracle/capture_golden.py run the reference's check_mesh_contains).
# Nothing is copied into the repository; needs cython + g++ + the Python/numpy headers of this image.
set -e
REF=${1:-/root/reference}
SRC=$REF/src/dev_fn/external/libmesh/triangle_hash.pyx
OUT=$(cd "$(dirname "$0")" && pwd)/_ref/libmesh
[ -f "$SRC" ] || { echo "reference not present: skipping oracle/_ref"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
NPINC=$(python3 -c 'import numpy; print(numpy.get_include())')
EXT=$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
python3 -m cython --cplus -3 -o "$OUT/triangle_hash.cpp" "$SRC"
g++ -O2 -shared -fPIC -std=c++14 -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I"$PYINC" -I"$NPINC" -o "$OUT/triangle_hash$EXT" "$OUT/triangle_hash.cpp"
rm -f "$OUT/triangle_hash.cpp"
echo "built $OUT/triangle_hash$EXT"
