export TAG=r05_final ROUND=r05
bash tools/gpu/run.sh final
