"""dctfhe -- MI355X-native homomorphic evaluation behind the DCT-CryptoNets operator surface.

Host-side mirror of the third-party boundary the reference calls:
compile_brevitas_qat_model / compile_torch_model / Configuration and the QuantizedModule they return.
All ciphertext arithmetic runs in libdctfhe.so (hand-written HIP, gfx950); there is no CPU path.
"""
