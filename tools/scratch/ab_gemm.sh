# A/B of variant libraries on the int4 GEMM shapes: default, then each argument
echo default; timeout 200 python tools/bench_shapes.py 2>&1 | grep -E "int4 linear"
for lib in "$@"; do echo "$lib"; FQHIP_LIB=$lib timeout 200 python tools/bench_shapes.py 2>&1 | grep -E "int4 linear"; done
