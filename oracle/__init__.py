"""CPU oracle for the fake-quant hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (llm-qat_amd/) never does.
"""
