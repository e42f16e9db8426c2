"""Player base classes (rlzero/mcts/player.py:5-57)."""


class Player(object):

    def __init__(self, player_id=0, player_name='') -> None:
        self.player_id = player_id
        self.player_name = player_name
        self.can_click = False

    def set_player_id(self, player_id):
        self.player_id = player_id

    def get_player_id(self):
        return self.player_id

    def get_player_name(self):
        return self.player_name

    def reset_player(self):
        raise NotImplementedError

    def get_action(self, game_env, **kwargs):
        raise NotImplementedError

    def __str__(self):
        return 'player'


class HumanPlayer(Player):
    """The import path of the reference's interactive player (rlzero/mcts/player.py:33-57) -- console I/O, out of scope here
    (SURVEY.md section 2): the class exists so that code naming it imports; it does not play."""

    def get_action(self, game_env, **kwargs):
        raise NotImplementedError('interactive play is not part of this package: use the reference\'s HumanPlayer')
