"""etl_amd — Python host mirror of the MI355X-native pgoutput decode stage (libetl_gfx950.so).

Importing the package sets ONE default of the HIP runtime for this process, if the caller has not set it: the number of hardware
queues. A decode context runs its work on up to seven HIP streams (two decode streams for consecutive ASYNC batches, the result
copies, the record-boundary scan, the control pre-pass, uploads, downloads) and ROCm multiplexes all streams of a process onto
GPU_MAX_HW_QUEUES hardware queues (4 by default). With a second context — or torch's own streams — in the process, the two decode
streams of a context end up on ONE hardware queue: consecutive batches then run one after the other instead of side by side
(measured: cfg3 chain 470 -> 350 GB/s, cfg2 without the no-control assertion 1 440 -> 740 GB/s), and a batch that waits on the device
for its predecessor's transaction state can sit in front of it in the queue until its bounded spin gives up (a 12 s kernel, then the
redo path). The variable is read when the HIP runtime initialises, i.e. at the first HIP call of the process: import this package (or
export the variable) before that. libetl_gfx950.so does the same from a load-time constructor for hosts that are not Python."""
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
