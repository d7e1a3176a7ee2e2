"""CPU oracle for the stain-normalisation hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker.  The product package
(``stainx_amd``) never imports this module and fails loudly when its HIP
library is missing.

Parity pinning: the restatement in ``stain_oracle.py`` is checked against
outputs of the real reference (stainx 0.1.4, ``backend="torch"`` on CPU,
imported from /root/reference/src in the build container) that are committed
as fixtures under ``tests/golden/`` together with the generating script
``tests/golden/make_golden.py``.
"""
