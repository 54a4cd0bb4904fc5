for g1 in 128 256 512; do echo "G1=$g1"; SOAP_ENGINE_G1=$g1 CONFIGS=only1 python scripts/small_perf.py 2>&1 | grep only; done
for nm in 65536 131072; do echo "NB_MEDIUM=$nm"; SOAP_ENGINE_NB_MEDIUM=$nm CONFIGS=only1,only2,all python scripts/small_perf.py 2>&1 | grep "only\|all"; done
