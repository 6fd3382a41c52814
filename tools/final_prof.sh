#!/bin/bash
# final profiles of the round (stats of all configs + pmc + nt/tn tables)
bash tools/profile_round.sh stats > gpurun_out/round_stats.log 2>&1
bash tools/profile_round.sh graph >> gpurun_out/round_stats.log 2>&1
tail -5 gpurun_out/round_stats.log
