#!/bin/bash
# everything the round's final profiles/ set needs, in one GPU call: bench lines, kernel traces, step breakdowns
SET=${1:-rXX}
bash tools/final_lines.sh $SET && bash tools/profile_b4.sh $SET && bash tools/profile_collect.sh $SET && echo all done
