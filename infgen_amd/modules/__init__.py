from .layers import AttentionLayer, FourierEmbedding, MLPEmbedding, MLPLayer
from .attr_tokenizer import Attr_Tokenizer
from .map_decoder import InfGenMapDecoder
from .agent_decoder import InfGenAgentDecoder
from .infgen_decoder import InfGenDecoder
from .token_processor import TokenProcessor, fetch_enterings, match_token_map

__all__ = ['TokenProcessor', 'fetch_enterings', 'match_token_map', 'AttentionLayer', 'FourierEmbedding', 'MLPEmbedding', 'MLPLayer', 'Attr_Tokenizer',
           'InfGenMapDecoder', 'InfGenAgentDecoder', 'InfGenDecoder']
