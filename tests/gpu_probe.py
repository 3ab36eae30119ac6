import sys, time, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
import pyoracle as O
t=time.time()
e = mumemto_amd.Engine(0)
docs = synth.pangenome(8, 400000, 0.005, seed=33)
e.set_docs(docs)
e.run()
print("n", e.text_length(), "rows", len(e.rows_mum()[0]), "stage_ms", e.stage_ms(), "wall", time.time()-t)
t=time.time()
want = O.run(docs)
print("oracle sec", time.time()-t, "equal", want.text()==e.output_text(), len(want.text()))
print(e.output_text()[:200])
