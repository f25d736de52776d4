"""TEST STAND-IN for the ``fairseq`` package (absent from the build image, no network to install it).

It reproduces what matters for the drop-in boundary: the registries of fairseq.models / fairseq.tasks /
fairseq.criterions REJECT classes that do not extend BaseFairseqModel / FairseqTask / FairseqCriterion and dataclasses
that do not extend FairseqDataclass, reject duplicate names, and ``fairseq.utils.import_user_module`` imports a
``--user-dir`` the way fairseq-train does.  Only tests/test_fairseq_boundary_cpu.py puts this directory on sys.path (in
a child process); nothing else in the repository can see it.
"""
__version__ = "0.12.2+standin"
from . import metrics, utils  # noqa: F401,E402
