"""alpha_omok_amd -- MI355X-native AlphaZero self-play engine for Omok.

Drop-in for the hot path of reinforcement-learning-kr/alpha_omok (ZeroAgent.get_pi /
main.self_play): batched MCTS over structure-of-arrays trees in HBM, hand-written HIP kernels
for gfx950, behind a C ABI (include/omok_hip.h).
"""
__version__ = "0.1.0"
