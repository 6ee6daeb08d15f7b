#!/bin/bash
# Builds tests/micro/variants/tp_nopad.so: the product library with the large-window role's first LDS layout (-DPVBA_TP_PAD=0 in ba_kernels.hip AND ba_solver.cpp --
# the host sizes the chunks with the same constants), linked with the product's other objects (run `make -C pvio_amd/csrc` first).  For tests/micro/tp_pad_ab.py.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/pvio_amd/csrc; T=$(mktemp -d); mkdir -p $R/tests/micro/variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -I$C -I$R/include -DPVBA_TP_PAD=0"
/opt/rocm/bin/hipcc $F -c $C/ba_kernels.hip -o $T/ba_kernels.o
/opt/rocm/bin/hipcc $F -x hip -c $C/ba_solver.cpp -o $T/ba_solver.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tests/micro/variants/tp_nopad.so $T/ba_kernels.o $T/ba_solver.o $C/klt.o $C/ba_comm.o $C/capi.o $C/preintegrator.o $C/sym_eig.o $C/sym_eig_avx2.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf $T; ls -la $R/tests/micro/variants/tp_nopad.so
