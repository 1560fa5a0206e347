#!/usr/bin/env python3
"""summarize a hipcc -Rpass-analysis=kernel-resource-usage stderr dump: python tools/dbg/kres_txt.py file [filter]"""
import re,sys,subprocess
t=open(sys.argv[1]).read(); flt=sys.argv[2] if len(sys.argv)>2 else ''
cur=None; v=None
for line in t.splitlines():
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().replace('void pz::','').split('(')[0]
    m=re.search(r' VGPRs: (\d+)',line)
    if m: v=m.group(1)
    m=re.search(r'ScratchSize \[bytes/lane\]: (\d+)',line)
    if m and cur and flt in cur: print('%-60s vgpr %s scratch %s' % (cur, v, m.group(1)))
