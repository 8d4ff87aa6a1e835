export TAG=r05_l
bash tools/gpu/run.sh env CMFREC_HIP_VH_MIN 513 385 321 258
