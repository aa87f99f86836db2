#!/bin/bash
# Build the product library and the measurement tools with absolute paths; stop on the first error.
# Tools that call the product through its C-ABI link libfeather_hip.so (rpath = its in-tree location).
set -e
R=/root/repo
make -s -j8 -C $R/feathercnn_amd/csrc
mkdir -p $R/tools/_build
for t in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -DFHIP_TIMELINE -I$R/include -I$R/feathercnn_amd/csrc -I$R/tools -I$R/tools/experiments $R/tools/$t.hip \
     -L$R/feathercnn_amd -lfeather_hip -Wl,-rpath,'$ORIGIN/../../feathercnn_amd' -o $R/tools/_build/$t
  echo "built tools/_build/$t"
done
