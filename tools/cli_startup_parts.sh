#!/bin/bash
t() { S=$(date +%s%N); "$@" > /dev/null 2>&1; E=$(date +%s%N); echo "$(( (E - S) / 1000000 )) ms  $*"; }
t python -c "pass"
t python -c "import numpy"
t python -c "import numpy, ctypes; l=ctypes.CDLL('normalisr_amd/libnormalisr_hip.so')"
t python -c "import numpy, ctypes; l=ctypes.CDLL('normalisr_amd/libnormalisr_hip.so'); print(l.nrm_device_count())"
t python -c "import numpy, ctypes; l=ctypes.CDLL('normalisr_amd/libnormalisr_hip.so'); l.nrm_set_device(0); import numpy as np; a=np.zeros(4); l.nrm_host_pin(a.ctypes.data, 32)"
t python -c "import torch"
t python -c "import torch; torch.zeros(1, device='cuda')"
t python -c "import normalisr_amd.__main__"
NRM_HOST_ENTRY=1 python -X importtime -c "import normalisr_amd.__main__" 2>&1 | sort -t'|' -k2 -n | tail -8
