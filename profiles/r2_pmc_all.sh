#!/bin/bash
# PMC passes for every bench configuration -> gpurun_out/r2pmc/<config>/pmc.json
cd $GRAFT_REPO_ROOT
bash profiles/r2_pmc.sh r2pmc/rgb 2 2048 && echo rgb ok
bash profiles/r2_pmc.sh r2pmc/perceptual 2 2048 --config perceptual && echo perceptual ok
bash profiles/r2_pmc.sh r2pmc/dither 2 2048 --config dither && echo dither ok
bash profiles/r2_pmc.sh r2pmc/images 1 2048 --config images && echo images ok
