#!/bin/bash
# build a comparison library with extra defines on conp_zn.hip: bash tools/build_variant.sh NAME "-DZN_TIMELINE" -> conp_amd/libconp_hip_NAME.so
set -e
C=$(dirname "$0")/../lammps-user-conp2_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c $C/conp_zn.hip -o $C/conp_zn_var_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $C/../conp_amd/libconp_hip_$1.so $C/conp_kernels.o $C/conp_inverse.o $C/conp_pppm.o $C/conp_rows.o $C/conp_tables.o $C/conp_zn_var_$1.o $C/conp_fix.o $C/conp_host.o
