"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's depth-marginalised DUN path (cambridge-mlg/DUN).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product package ``dun_amd`` never does.
"""
