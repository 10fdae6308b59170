#!/bin/bash
# The CPU fiber-emulator build of the kernel sources with AddressSanitizer: every global / LDS access of every emulated kernel
# launch is bounds-checked against the real allocations (torch's CPU tensors are ordinary heap blocks under the preloaded runtime).
#   tools/build_emu_asan.sh            -> $VTX_ASAN_DIR/libvirtex_amd_emu_asan.so (default /tmp/virtex_amd_emu_asan: 220 MB of
#                                         objects that must not travel with the repo snapshot; ~5-8 min on 8 cores)
#   RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
#   HIPEMU_THREADS=1 HIPEMU_EXACT_LDS=1 LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 \
#     VTX_EMU_LIB=/tmp/virtex_amd_emu_asan/libvirtex_amd_emu_asan.so python -m pytest tests/test_kernels.py -q -m "not gpu"
# Static __shared__ arrays are plain statics in this build (red zones; hence ONE worker thread), dynamic LDS is a heap block of exactly
# the requested size per launch (HIPEMU_EXACT_LDS).  The fiber runtime itself (hand-written context switches, tests/hipemu/hipemu.cpp) stays uninstrumented.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); CL=${VTX_HOST_CLANG:-/opt/rocm/lib/llvm/bin/clang++}; D=${VTX_ASAN_DIR:-/tmp/virtex_amd_emu_asan}; O=$D/obj
mkdir -p $O
FL="-x c++ -O1 -g -std=c++17 -fPIC -pthread -DHIPEMU=1 -I $R/tests/hipemu/include -I $R/include -Wno-unused-value -Wno-unknown-pragmas -Wno-pass-failed -ffp-contract=off -fsanitize=address -fno-omit-frame-pointer -DHIPEMU_STATIC_LDS"
ls $R/virtex_amd/csrc/*.hip | xargs -P ${JOBS:-8} -I{} bash -c "b=\$(basename {} .hip); $CL $FL -c {} -o $O/\$b.o"
$CL -x c++ -O2 -g -std=c++17 -fPIC -pthread -DHIPEMU=1 -I $R/tests/hipemu/include -c $R/tests/hipemu/hipemu.cpp -o $O/hipemu_rt.o
$CL -shared -fsanitize=address -shared-libasan -pthread $O/*.o -o $D/libvirtex_amd_emu_asan.so
echo $D/libvirtex_amd_emu_asan.so
