# Round 6: the whole GPU suite under every remaining route switch (tools/alt_modes.sh), poison fixture on.  One gpurun call.
set -u
mkdir -p gpurun_out/r06
bash tools/alt_modes.sh > gpurun_out/r06/alt_modes.txt 2>&1; cat gpurun_out/r06/alt_modes.txt
