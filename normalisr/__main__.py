import sys

from normalisr_amd.__main__ import main

sys.exit(main())
