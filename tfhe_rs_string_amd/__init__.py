"""tfhe_rs_string_amd -- MI355X-native keyswitch + programmable-bootstrap engine for the
tfhe-rs string layer (drop-in for the KS-PBS hot path of M-Bln/tfhe-rs-string).

Only what the path needs: csrc/ (HIP kernels + the C-ABI of include/tfhe_hip.h) and the host-side
mirror of the reference's shortint / integer / FheString interface for that path.
"""
from ._native import (CLASSIC_KS_PBS, classic_params, PARAM_MESSAGE_2_CARRY_2_KS_PBS, PARAM_MESSAGE_2_CARRY_2_PBS_KS, PARAM_MESSAGE_2_CARRY_2_TEST,  # noqa: F401
                      PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS, PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS,
                      Params, TfheHipError)
from .engine import (Context, PinnedArray, ServerKey, compressed_server_key_info, csprng_words, deserialize_ciphertext,  # noqa: F401
                     decompress_ciphertexts, deserialize_compressed_ciphertext, generate_lookup_table, serialize_ciphertext,
                     serialize_compressed_ciphertext, serialize_compressed_server_key, serialize_keyswitch_key, serialize_bootstrap_key,
                     deserialize_keyswitch_key, deserialize_bootstrap_key, exact_external_product_batch)
from .strings import HostKey, IntegerServerKey, ShortintServerKey, StringServerKey, ResultGather, column_sharded, contains_sharded, shard_range  # noqa: F401,E402
