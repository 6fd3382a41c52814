#!/bin/bash
# round-3 loop: GPU suite, bf16 parity tools, headline bench -> gpurun_out/$1/
out=gpurun_out/${1:-r3}
mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
tail -15 $out/pytest.txt
PCB_CENTRE=0 python tools/bf16_parity.py > $out/parity_plain.txt 2>&1
python tools/bf16_parity.py > $out/parity.txt 2>&1
python bench.py --no-extras > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "exec", d["config"]["exec"], "roofline", d["roofline"]["frac"])
PY
