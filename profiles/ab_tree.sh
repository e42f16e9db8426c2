# same-box A / B of two builds of the library on the legs the tree step matters for: M sims/s (ms per move)
#   REF=profiles/tmp_prof/lib_head.so bash profiles/ab_tree.sh
F="--no-cpu-baseline --no-fill --no-configs --no-games-leg --regions 1"
run() {  # label, flags
    l=$1; shift
    python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LIB $l:', round(d['value']/1e6, 3), d['ms_per_step'])"
}
for rep in 1 2 3; do
for LIB in ref new; do
    if [ $LIB = ref ]; then export RZ_HIP_LIBRARY=$PWD/$REF; else unset RZ_HIP_LIBRARY; fi
    run headline --steps 8 --warmup 3
    run puct --steps 6 --warmup 3 --score-mode puct
    run c2 --steps 8 --warmup 8 --board 9 --playouts 200 --games 64
    run c3 --steps 6 --warmup 6 --game connect4 --playouts 400 --games 512
    run c1x16 --steps 9 --warmup 20 --board 3 --playouts 25 --games 16
done
done
