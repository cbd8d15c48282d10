// MenciusNative.scala -- the Mencius half of the reference-side binding (source only, like Native.scala: no JDK /
// scalac in this image).  Drop into jvm/src/main/scala/frankenpaxos/gpu/ next to Native.scala.
//
// `GpuMenciusProxyLeader` stands where a mencius.ProxyLeader stands (mencius/ProxyLeaderMain.scala).  Leaders of
// EVERY leader group keep sending it Phase2a and Phase2aNoopRange exactly as today
// (mencius/Leader.scala:342-345, 455); its `receive` only enqueues; one zero-delay Transport timer per burst
// flushes the queue through TWO native calls -- the commands (Native.phase2Fused = mencius.ProxyLeader.handlePhase2a
// + every mencius.Acceptor.handlePhase2a + mencius.ProxyLeader.handlePhase2b, mencius/ProxyLeader.scala:216-253,
// 305-353, mencius/Acceptor.scala:202-235) and the noop ranges (Native.noopRangesFused = the *NoopRange handlers,
// mencius/ProxyLeader.scala:255-303, 355-411, mencius/Acceptor.scala:237-291) -- and then `send`s what the Scala
// handlers would have sent: Chosen / ChosenNoopRange to every replica, Nack(round) to
// leaders(slotSystem.leader(slot))(roundSystem.leader(round)) (mencius/Acceptor.scala:215-217).
//
// The context holds every leader group's acceptor groups (fpx_config: num_leader_groups = numLeaderGroups,
// num_groups = acceptor groups per leader group; slot -> leader group slot % L, acceptor group (slot / L) % A,
// mencius/ProxyLeader.scala:169-176, 231-234).  Its rows are leader-group-major in HBM (include/fpx.h,
// FPX_F_SLOT_MAJOR_ROWS), so the tick hands the commands over AS THE LEADER GROUPS' BATCHES BACK TO BACK, each in slot
// order: the queue is a bucket per leader group, which is also the order in which one leader's messages arrive.
// (A single slot-ordered batch is regrouped by the kernel itself at a small cost: profiles/r03_cfg5.md.)
//
// Scope: the log window is [0, numSlots) -- the windowing / recycling of GpuPhase2Engine (Native.scala) applies
// unchanged (recycleSlots takes slots; with leader-group-major rows a window must be a multiple of numLeaderGroups)
// and is left out here to keep the seam readable.  HighWatermark messages are forwarded as in the reference.
package frankenpaxos.gpu

import frankenpaxos.Actor
import frankenpaxos.Chan
import frankenpaxos.Logger
import frankenpaxos.mencius._
import frankenpaxos.roundsystem.RoundSystem
import scala.collection.mutable

class GpuMenciusProxyLeader[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    numSlots: Int = 1 << 22
) extends Actor(address, transport, logger) {
  override type InboundMessage = ProxyLeaderInbound
  override val serializer = ProxyLeaderInboundSerializer

  config.checkValid()
  private val L = config.numLeaderGroups
  private val A = config.acceptorAddresses(0).size            // acceptor groups per leader group
  private val R = config.acceptorAddresses(0)(0).size         // acceptors per group (2f + 1)
  logger.check(numSlots % L == 0)
  // fpx_config as fpx_jni.c reads it: slots, replicas, groups, leader groups, f, quorum kind, grid rows / cols,
  // leaders per group, ballot model (ACCEPTOR: noop ranges act on the acceptor's round), tally ways, replica base /
  // total, device, flags
  private val handle = Native.create(
    Array(numSlots, R, A, L, config.f, /*THRESHOLD*/ 0, 0, 0, config.leaderAddresses(0).size, /*ACCEPTOR*/ 0, 4, 0, 0, 0, 0))
  if (handle < 0) Native.check((-handle).toInt, logger)

  private val slotSystem = new RoundSystem.ClassicRoundRobin(L)
  private val roundSystem = new RoundSystem.ClassicRoundRobin(config.leaderAddresses(0).size)
  private val leaders = for (group <- config.leaderAddresses)
    yield for (a <- group) yield chan[Leader[Transport]](a, Leader.serializer)
  private val replicas = for (a <- config.replicaAddresses) yield chan[Replica[Transport]](a, Replica.serializer)

  // value ids: the int32 the GPU carries stands for the CommandBatchOrNoop kept here (Noop = -1), as in GpuPhase2Engine
  private val values = mutable.ArrayBuffer[CommandBatchOrNoop]()
  private def intern(v: CommandBatchOrNoop): Int = if (v.value.isNoop) -1 else { values += v; values.size - 1 }
  private def valueOf(id: Int): CommandBatchOrNoop = if (id < 0) CommandBatchOrNoop().withNoop(Noop()) else values(id)

  // one bucket of commands per leader group; the noop ranges of the burst in arrival order
  private val pending = Array.fill(L)(mutable.Buffer[Phase2a]())
  private val pendingRanges = mutable.Buffer[Phase2aNoopRange]()
  private var queued = 0
  private val tick = timer("gpuMenciusTick", java.time.Duration.ZERO, () => flushTick())

  override def receive(src: Transport#Address, inbound: ProxyLeaderInbound): Unit = {
    import ProxyLeaderInbound.Request
    inbound.request match {
      case Request.Phase2A(p) =>
        if (queued == 0) tick.start()
        pending(slotSystem.leader(p.slot)) += p
        queued += 1
      case Request.Phase2ANoopRange(p) =>
        if (queued == 0) tick.start()
        pendingRanges += p
        queued += 1
      case Request.HighWatermark(h) =>                       // mencius/ProxyLeader.scala:207-214
        for (group <- leaders; leader <- group) leader.send(LeaderInbound().withHighWatermark(h))
      case Request.Phase2B(_) | Request.Phase2BNoopRange(_) =>
        logger.fatal("GpuMenciusProxyLeader tallies on the device; it never receives Phase2b messages.")
      case Request.Empty =>
        logger.fatal("Empty ProxyLeaderInbound encountered.")
    }
  }

  private def nack(slot: Int, round: Int, acceptorsRound: Int): Unit =
    leaders(slotSystem.leader(slot))(roundSystem.leader(round)).send(LeaderInbound().withNack(Nack(round = acceptorsRound)))

  private def flushTick(): Unit = {
    // ---- commands: the leader groups' batches back to back, each sorted by slot.  A slot that appears twice in the
    // burst (a re-proposal in a higher round) must not share a device run with its first message: the library
    // splits such a batch into runs itself (host entry points), in message order.  Regrouping the burst is a
    // reordering of messages that were in flight together -- something the asynchronous network may do anyway; one
    // leader's own messages (increasing slots of its group) keep their order, and sortBy is stable for equal slots
    val now = pending.flatMap(_.sortBy(_.slot)).toArray
    pending.foreach(_.clear())
    val n = now.length
    if (n > 0) {
      val slot = now.map(_.slot); val round = now.map(_.round); val value = now.map(p => intern(p.commandBatchOrNoop))
      val chosen = new Array[Byte](n); val cr = new Array[Int](n); val cv = new Array[Int](n); val nr = new Array[Int](n)
      // dense delivery; a thrifty deployment passes a random quorumSize of the slot's group per message instead
      // (mencius/ProxyLeader.scala:236: rand.shuffle(group).take(config.quorumSize))
      Native.check(Native.phase2Fused(handle, n, slot, round, value, null, chosen, cr, cv, nr), logger)
      for (i <- 0 until n) {
        if (chosen(i) != 0)                                   // mencius/ProxyLeader.scala:338-352
          replicas.foreach(_.send(ReplicaInbound().withChosen(Chosen(slot = slot(i), commandBatchOrNoop = valueOf(cv(i))))))
        if (nr(i) >= 0) nack(slot(i), round(i), nr(i))        // mencius/Acceptor.scala:208-219
      }
    }
    // ---- noop ranges: one fused launch for all of them (up to 4096 of them walk the chain in one workgroup)
    val m = pendingRanges.size
    if (m > 0) {
      val start = pendingRanges.map(_.slotStartInclusive).toArray
      val end = pendingRanges.map(_.slotEndExclusive).toArray
      val round = pendingRanges.map(_.round).toArray
      val isNew = new Array[Byte](m); val chosen = new Array[Byte](m); val nr = new Array[Int](m)
      val votes = new Array[Long](m * A * 4); val nacks = new Array[Long](m * A * 4)
      Native.check(Native.noopRangesFused(handle, m, A, start, end, round, null, votes, nacks, nr, isNew, chosen), logger)
      for (i <- 0 until m) {
        if (chosen(i) != 0)                                   // mencius/ProxyLeader.scala:395-407
          replicas.foreach(_.send(ReplicaInbound().withChosenNoopRange(
            ChosenNoopRange(slotStartInclusive = start(i), slotEndExclusive = end(i)))))
        if (nr(i) >= 0) nack(start(i), round(i), nr(i))       // mencius/Acceptor.scala:245-256
      }
      pendingRanges.clear()
    }
    queued = 0
  }

  def close(): Unit = Native.check(Native.destroy(handle), logger)
}
