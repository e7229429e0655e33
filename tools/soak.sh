#!/bin/bash
# long replayed training on four fixed batches (an overfitting run: the radial MLPs sharpen): step time and knot-table vetoes
run() { echo -n "$1: "; env $1 python3 bench.py --no-cpu-baseline --steps ${STEPS:-1000} --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['median'], d['ms_per_step_repeats']['max'], 'loss', d['config']['final_loss'], 'recaptures', d['config']['knot_table_recaptures'], 'knots', d['config']['knot_table_knots'])"; }
run "E3K_X=0"                            # the default: a tripped guard refines the tables
run "E3K_RADIAL_KNOTS_MAX=512"           # no finer table allowed: the guard switches the MLP's table off
run "E3K_RADIAL_TABLE_TOL_COL=2e-5"      # round 5's per-column tolerance
run "E3K_RADIAL_KNOTS=1024"              # fine from the start
