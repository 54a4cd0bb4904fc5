# predictor on / off over seeds and workloads (scripts/ab.py)
for s in ${SEEDS:-1234 5}; do echo "seed $s"; SEED=$s timeout -k 10 300 python scripts/ab.py SOAP_ENGINE_PREDICT=0 - 2>&1 | grep -v "^\[soap\|amdgpu.ids" | cut -c1-330; done
for w in ${WLS:-giant c2}; do echo "workload $w"; WL=$w timeout -k 10 300 python scripts/ab.py SOAP_ENGINE_PREDICT=0 - 2>&1 | grep -v "^\[soap\|amdgpu.ids" | cut -c1-330; done
