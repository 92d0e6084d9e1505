import sys, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools/ab')
import numpy as np
import oracle.oracle as oo
oo.build = lambda force=False: '/root/repo/tools/ab/liboracle_rule.so'
from oracle.oracle import OracleQp, default_opts, lib
import rule_eval
