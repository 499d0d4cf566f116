# Quadrant masks against plain lists.  At the time of profiles/r04_qmask.txt the masks were the VARIANT (-DTS2D_QMASK) and the
# plain lists the product; since then the masks are the default and the plain lists are the variant:
#     tools/build_flag_variant.sh plain -DTS2D_NO_QMASK          (in the build container: tools/bin/libts2d_plain.so + libts2d_lab_plain.so)
#     gpurun -- 'bash tools/r04_qmask.sh'
# runs the 2D parity suites through the variant (with its own lab library for the state reader), then product and variant alternating on one box
# (bench.py's per-kernel HIP-event averages).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/tools/bin/libts2d_plain.so
if [ "$1" != "bench" ]; then
echo "== parity through the variant"
TS2D_LIBRARY_PATH=$V TS2D_LAB_LIBRARY_PATH=$R/tools/bin/libts2d_lab_plain.so timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_reference_gpu.py tests/test_speculative_forward_gpu.py -x -q 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_qmask_pytest.log
grep -n "^E  \|Fatal\|passed\|failed\|^FAILED\|^tests.*Error" gpurun_out/r04_qmask_pytest.log | head -30
fi
if [ "$1" != "parity" ]; then
echo "== product / variant alternating"
bash tools/ab_bench.sh $V 2>&1 | grep -v amdgpu.ids
fi
